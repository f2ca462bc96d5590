"""The EmeiEnv-shaped Python surface driven the way the reference's own tests and callers drive it."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import emei_amd  # noqa: E402


@pytest.mark.parametrize("name,cls", [("swingup", emei_amd.CartPoleSwingUpEnv), ("balancing", emei_amd.CartPoleBalancingEnv)])
@pytest.mark.parametrize("fr", [1, 4])
def test_single_env_reproduces_reference_trajectory(cartpole_golden, name, cls, fr):
    """BASELINE configs[0] through the gym-style API: reset(seed) + 1000 x step(int)."""
    g = cartpole_golden
    tag = f"traj_{name}_fr{fr}_seed0"
    env = cls(freq_rate=fr)
    obs, info = env.reset(seed=0)
    assert info == {} and obs.dtype == np.float64 and np.array_equal(obs, g[tag + "_states"][0])
    T = len(g[tag + "_actions"])
    assert T == 1000  # all of config 1: CartPoleSwingUp-v0's max_episode_steps (register_env.py:19-23)
    for t in range(T):
        obs, reward, terminal, truncated, info = env.step(int(g[tag + "_actions"][t]))
        assert isinstance(reward, np.float64) and isinstance(terminal, np.bool_) and truncated is False and info == {}
        # the float64 state follows the reference to ~1e-9 over the first 300 steps; the chaotic pendulum
        # amplifies the last-bit differences of sincos afterwards: north-star tolerance 1e-5 over all 1000
        if t < 300:
            assert rel_err(obs, g[tag + "_states"][t + 1], floor=1e-30) <= 1e-9, t
        else:
            assert rel_err(obs, g[tag + "_states"][t + 1]) <= 1e-5, t
        assert abs(reward - g[tag + "_reward"][t]) <= 1e-6 and bool(terminal) == bool(g[tag + "_terminal"][t]), t


def test_reference_behaviour_tests_cartpole():
    """test_cartpole.py:14-35: both envs eventually return terminal=True under action_space.sample()."""
    for cls in (emei_amd.CartPoleBalancingEnv, emei_amd.CartPoleSwingUpEnv):
        env = cls()
        obs, info = env.reset(seed=3)
        env.action_space.seed(3)
        for _ in range(5000):
            obs, reward, terminal, truncated, info = env.step(env.action_space.sample())
            if terminal:
                break
        assert terminal


def test_step_before_reset_and_bad_action():
    env = emei_amd.CartPoleSwingUpEnv()
    with pytest.raises(AssertionError, match="Call reset before using step method"):
        env.step(0)
    env.reset(seed=0)
    with pytest.raises(AssertionError, match="invalid"):
        env.step(2)  # not in Discrete(2): base_control.py:65-66
    with pytest.raises(AssertionError, match="invalid"):
        env.step(0.5)


def test_make_applies_time_limit():
    env = emei_amd.make("CartPoleBalancing-v0", num_envs=64)
    env.reset(seed=0)
    acts = torch.randint(0, 2, (500, 64), device=env.engine.device, dtype=torch.uint8)
    obs, rew, term, trunc = env.rollout(acts)
    assert bool(trunc[-1].all()) and not bool(trunc[:-1].any())  # register_env.py:14-18: 500 steps


def test_vectorised_env_and_batch_functions(cartpole_golden):
    g = cartpole_golden
    env = emei_amd.CartPoleSwingUpEnv(num_envs=128)
    obs, _ = env.reset(seed=0)
    want = np.random.default_rng(0).uniform(-0.05, 0.05, (128, 4))
    want[:, 2] += np.pi
    assert np.array_equal(obs.cpu().numpy(), want)  # same PCG64 stream as the reference's reset(seed=0)
    o, r, term, trunc, info = env.step(torch.ones(128, dtype=torch.int64, device=obs.device))
    assert o.shape == (128, 4) and r.shape == (128,) and term.dtype == torch.bool and not bool(trunc.any())
    # numpy in -> numpy [B,1] out, like the reference; torch in -> torch out
    ob = g["batch_swingup_obs"]
    r_np, t_np = env.get_batch_reward(ob), env.get_batch_terminal(ob)
    assert r_np.shape == (len(ob), 1) and r_np.dtype == np.float64 and t_np.dtype == np.bool_
    assert rel_err(r_np, g["batch_swingup_reward"]) <= 1e-5  # every row: float64 NumPy rows reach the kernel unnarrowed
    assert np.array_equal(t_np, g["batch_swingup_terminal"])
    assert isinstance(env.get_batch_reward(torch.as_tensor(ob, device=obs.device)), torch.Tensor)
    assert np.array_equal(emei_amd.CartPoleBalancingEnv().get_batch_reward(ob), np.ones((len(ob), 1)))
    # get_batch_next_obs needs a frozen env (core.py:190-193) and leaves the env's own state alone
    with pytest.raises(AssertionError):
        env.get_batch_next_obs(ob, action=np.zeros(len(ob), np.int64))
    before = env.engine.get_state().clone()
    env.freeze()
    nxt = env.get_batch_next_obs(g["onestep_swingup_state"][:64], action=g["onestep_swingup_action"][:64])
    env.unfreeze()
    assert torch.equal(env.engine.get_state(), before)
    assert rel_err(nxt, g["onestep_swingup_fr1_dt0.02_next"][:64]) <= 1e-5


def test_inverted_pendulum_env_api():
    env = emei_amd.BoundaryInvertedPendulumBalancingEnv()
    np.random.seed(0)
    obs, _ = env.reset()
    assert obs.shape == (4,) and abs(obs[0] - obs[1]) < 1e-15 and obs[2] == obs[3] and abs(obs[0]) < 0.05  # B=1 noise quirk (theta went through the wrap)
    for _ in range(2000):
        obs, reward, terminal, truncated, info = env.step(env.action_space.sample())
        if terminal:
            break
    assert terminal and reward == 1.0
    with pytest.raises(ValueError):
        env.step(np.zeros(2, np.float32))


def test_reset_seed_reaches_device_generator():
    """reset(seed=) also keys the device generator that draws every auto-reset episode's initial state: two seeds
    give different episode-1 initial observations, the same seed the same, and successive un-seeded resets differ
    (the host-reset path used to leave the key at 0)."""
    env = emei_amd.CartPoleBalancingEnv(num_envs=64, auto_reset=True, max_episode_steps=5)
    acts = torch.zeros((5, 64), dtype=torch.uint8, device=env.engine.device)

    def episode1_init(seed):
        env.reset(seed=seed)
        obs, rew, term, trunc = env.rollout(acts)
        assert bool((trunc[-1] | term[-1]).all())  # TimeLimit 5: every env was re-initialised on the device
        return env.engine.get_obs().clone()

    a, b, a2 = episode1_init(1), episode1_init(2), episode1_init(1)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    env.reset()  # un-seeded reset after reset(seed=1): a fresh key, np_random untouched
    env.rollout(acts)
    assert not torch.equal(env.engine.get_obs(), a)


def test_output_buffers_are_validated():
    """Engine.step / rollout(out=...): the ABI takes raw pointers without sizes, so a wrong buffer is refused on the host."""
    from emei_amd.engine import Engine

    eng = Engine("CartPoleSwingUp", 128)
    eng.reset(0)
    a = torch.zeros(128, dtype=torch.uint8, device=eng.device)
    good = eng.alloc_outputs(None)
    eng.step(a, out=good)
    eng.step(a, out=good)
    assert not hasattr(eng, "_out_ok")  # nothing of the caller's buffers is retained between calls
    assert eng.last_kernel() != 0  # emei_step goes through emei_rollout: the getter covers the step path too
    obs, rew, done = good
    for bad in ((obs[:64], rew, done), (obs, rew.double(), done), (obs, rew, done.to(torch.int32)), (obs.cpu(), rew, done),
                (torch.empty((128, 8), dtype=torch.float32, device=eng.device)[:, ::2], rew, done), (obs, rew), None):
        if bad is None:
            continue
        with pytest.raises(ValueError):
            eng.step(a, out=bad)
    acts = torch.zeros((16, 128), dtype=torch.uint8, device=eng.device)
    with pytest.raises(ValueError):
        eng.rollout(acts, out=good)  # step-shaped buffers for a 16-step rollout
    eng.rollout(acts, out=eng.alloc_outputs(16))
    # validated on every call: a list element replaced, or a tensor of an accepted tuple shrunk in place, is seen
    lst = list(eng.alloc_outputs(None))
    eng.step(a, out=lst)
    lst[0] = obs[:64]
    with pytest.raises(ValueError):
        eng.step(a, out=lst)
    tup = eng.alloc_outputs(None)
    eng.step(a, out=tup)
    tup[1].resize_(64)
    with pytest.raises(ValueError):
        eng.step(a, out=tup)


def test_freeze_reset_unfreeze_keeps_the_auto_reset_episodes():
    """the key of the device reset generator is part of the frozen snapshot.  freeze -> reset(seed=other) ->
    unfreeze -> auto-reset rollout must reproduce the rollout of an env that was never re-seeded."""
    import emei_amd

    acts = torch.as_tensor(np.random.default_rng(5).integers(2, size=(300, 256)), device="cuda")

    def run(reseed):
        env = emei_amd.CartPoleBalancingEnv(num_envs=256, auto_reset=True, max_episode_steps=500)
        env.reset(seed=11)
        env.freeze()
        if reseed:
            env.reset(seed=99)
            env.rollout(acts[:7])
        env.unfreeze()
        return env.rollout(acts)

    a, b = run(False), run(True)
    assert int((a[2] | a[3]).sum()) > 256  # every env went through several auto-resets
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("name,n", [("CartPoleSwingUp", 1), ("CartPoleSwingUp", 5), ("BoundaryInvertedPendulumSwingUp", 1), ("HopperRunning", 1), ("HalfCheetahRunning", 3)])
def test_step_host_equals_step_plus_get_obs(name, n):
    """emei_step_host (round 4): the single-launch path with its polled completion word (one env of the 4-state family) and the
    step + emei_get_obs + synchronise path (several envs, the bodies) return what emei_step / emei_get_obs return, bit for bit,
    over 40 steps with auto-reset and a TimeLimit of 7 (resets inside the window: the float64 observation is then the NEW episode's)."""
    from emei_amd.engine import Engine

    rng = np.random.default_rng(9)
    kw = dict(max_episode_steps=7, seed=4, init_noise=5e-3 if "CartPole" not in name else 0.0)
    a_eng, b_eng = Engine(name, n, **kw), Engine(name, n, **kw)
    a_eng.reset(4), b_eng.reset(4)
    for t in range(40):
        if a_eng.act_dim == 0:
            act = rng.integers(0, 2, n)
            dev = torch.as_tensor(act, device=b_eng.device)
        else:
            act = rng.uniform(-1, 1, (n,) if a_eng.act_dim == 1 else (n, a_eng.act_dim)).astype(np.float32)
            dev = torch.as_tensor(act, device=b_eng.device)
        o64, o32, rew, done = a_eng.step_host(act, auto_reset=True)
        obs, r, d = b_eng.step(dev, auto_reset=True)
        assert np.array_equal(o32, obs.cpu().numpy()) and np.array_equal(rew, r.cpu().numpy()) and np.array_equal(done, d.cpu().numpy())
        assert np.array_equal(o64, b_eng.get_obs().cpu().numpy())
    assert (np.asarray(done) >= 0).all()


def test_step_host_after_close_raises_instead_of_faulting():
    """`Engine.step_host` pre-binds the raw handle into a ctypes call; after `close()` that call held a dangling
    `emei_env*`.  close() drops it, a later step_host raises."""
    from emei_amd import _lib
    from emei_amd.engine import Engine

    eng = Engine("CartPoleSwingUp", 1)
    eng.reset(0)
    eng.step_host(1)
    eng.close()
    assert eng._host_io is None and eng._host_bufs is None
    with pytest.raises(_lib.EmeiHipError, match="closed"):
        eng.step_host(1)
    eng.close()  # idempotent
