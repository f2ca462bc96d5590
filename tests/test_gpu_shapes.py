"""Ragged and boundary shapes: every (n_envs, n_steps) combination must give the same numbers whichever
kernel the dispatch picks (staged: n % 64 == 0 and T >= 16; generic otherwise; T = 1 is emei_step)."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


def _actions(env, T, N, dev, dtype=None):
    if env.startswith("CartPole"):
        return torch.randint(0, 2, (T, N), device=dev).to(dtype or torch.uint8)
    if env == "HalfCheetahRunning":
        return (torch.rand((T, N, 6), device=dev) * 2 - 1).float()
    return (torch.rand((T, N), device=dev) * 6 - 3).float()


@pytest.mark.parametrize("env", ["CartPoleSwingUp", "CartPoleBalancing", "ReboundInvertedPendulumSwingUp", "BoundaryInvertedPendulumBalancing"])
@pytest.mark.parametrize("N", [1, 63, 64, 65, 1000, 1024])
@pytest.mark.parametrize("T", [1, 15, 16, 17, 48])
def test_rollout_equals_steps_every_shape(env, N, T):
    torch.manual_seed(N * 100 + T)
    a = _engine(env, N, freq_rate=3, max_episode_steps=11, seed=7, init_noise=5e-3)
    b = _engine(env, N, freq_rate=3, max_episode_steps=11, seed=7, init_noise=5e-3)
    a.reset(7)
    b.reset(7)
    acts = _actions(env, T, N, a.device)
    obs, rew, done = a.rollout(acts, auto_reset=True)
    for t in range(T):
        o, r, d = b.step(acts[t], auto_reset=True)
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t]), (t,)
    assert torch.equal(a.get_state(), b.get_state())
    assert torch.equal(a.compact_done(), torch.nonzero(done[-1]).flatten().to(torch.int32))
    sa, ea = a.get_counters()
    sb, eb = b.get_counters()
    assert torch.equal(sa, sb) and torch.equal(ea, eb)
    if T > 11:
        assert bool((done != 0).any()) and int(ea.max()) >= 1  # TimeLimit 11 (or an earlier terminal) ended an episode


@pytest.mark.parametrize("N,T", [(1, 1), (63, 5), (64, 16), (130, 17)])
def test_cheetah_shapes(N, T):
    torch.manual_seed(N + T)
    a = _engine("HalfCheetahRunning", N, freq_rate=2, real_time_scale=0.002, max_episode_steps=6, seed=3, init_noise=0.1)
    b = _engine("HalfCheetahRunning", N, freq_rate=2, real_time_scale=0.002, max_episode_steps=6, seed=3, init_noise=0.1)
    a.reset(3)
    b.reset(3)
    acts = _actions("HalfCheetahRunning", T, N, a.device)
    obs, rew, done = a.rollout(acts, auto_reset=True)
    for t in range(T):
        o, r, d = b.step(acts[t], auto_reset=True)
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t])
    assert torch.equal(a.get_state(), b.get_state())


@pytest.mark.parametrize("N", [1, 3, 65])
@pytest.mark.parametrize("precision", ["f32", "ref"])
def test_cheetah_chunking_does_not_depend_on_the_padding_lanes(N, precision):
    """Found by tools/stress.py in round 4: the padding lanes of a ragged wave run the wave's arithmetic on a state of their own.
    From the zero state a padding cheetah stood on both feet and fell differently in a fused rollout than in the same rollout
    cut into launches — and whether a real env with three row blocks found a wave-mate to borrow a constraint slot from
    (cheetah_model.h: `donor`), i.e. which of two solvers it ran, depended on the chunking.  Padding lanes are parked in the
    air now (Body::park).  Strong pushes, short episodes, eight action draws: the old library failed this in float32 for N = 1 and 3."""
    kw = dict(freq_rate=2, precision=precision, seed=179079704, max_episode_steps=20, env_index_offset=984416, real_time_scale=0.002,
              integrator="euler", init_noise=0.1, noise_layout="shared")
    T = 64
    for draw in range(8):
        torch.manual_seed(1000 * N + draw)
        a, b = _engine("HalfCheetahRunning", N, **kw), _engine("HalfCheetahRunning", N, **kw)
        a.reset(kw["seed"])
        b.reset(kw["seed"])
        acts = (torch.rand((T, N, 6), device=a.device) * 2.4 - 1.2).float()
        obs, rew, done = a.rollout(acts, auto_reset=True)
        cuts = [0, 7 + draw, 30 + 2 * draw, T]
        parts = [b.rollout(acts[lo:hi].contiguous(), auto_reset=True) for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert torch.equal(torch.cat([p[0] for p in parts]), obs), (draw,)
        assert torch.equal(torch.cat([p[2] for p in parts]), done) and torch.equal(a.get_state(), b.get_state())
        a.close()
        b.close()


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int32, torch.int64])
def test_action_dtypes_agree(dtype):
    N, T = 256, 40
    base = torch.randint(0, 2, (T, N), device="cuda")
    outs = []
    for dt in (torch.uint8, dtype):
        e = _engine("CartPoleSwingUp", N, seed=1)
        e.reset(1)
        outs.append(e.rollout(base.to(dt).contiguous()))
    assert all(torch.equal(x, y) for x, y in zip(*outs))


def test_null_outputs_are_skipped():
    """Any output pointer may be NULL (include/emei_hip.h): state still advances identically."""
    import ctypes as C

    from emei_amd import _lib as L
    from emei_amd.engine import _ptr, _stream

    N, T = 128, 32
    a, b = _engine("CartPoleSwingUp", N, seed=2), _engine("CartPoleSwingUp", N, seed=2)
    a.reset(2)
    b.reset(2)
    acts = torch.randint(0, 2, (T, N), device=a.device, dtype=torch.uint8)
    obs, rew, done = a.rollout(acts)
    rew_only = torch.empty((T, N), dtype=torch.float32, device=a.device)
    L.check(L.lib().emei_rollout(b._h, T, _ptr(acts), L.ACT_U8, C.c_void_p(0), _ptr(rew_only), C.c_void_p(0), 0, _stream()))
    assert torch.equal(rew_only, rew) and torch.equal(a.get_state(), b.get_state())
