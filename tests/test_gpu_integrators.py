"""mujoco_env.py:70-79,98-104 on the device: the three integrators and the per-substep observation
noise for every MuJoCo-backed body, against the oracle's restatement (oracle/integrators.h)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

INTEGRATORS = ["euler", "semi_implicit_euler", "rk4"]


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


def _case(body, rng, n):
    """(env name, state [n, sd], action [n(, ad)], oracle step fn -> next_state, obs-like, reward, terminal)"""
    from oracle import oracle as O

    if body == "ip":
        s = np.column_stack([rng.uniform(-2.2, 2.2, n), rng.uniform(-3.5, 3.5, n), rng.normal(0, 2, n), rng.normal(0, 4, n)])
        a = rng.uniform(-3.5, 3.5, (n, 1))
        f = lambda st, ac, fr, dt, o: (lambda r: (r[0], r[1], r[2], r[3]))(O.ip_step("boundary_swingup", st, ac[:, 0], fr, dt, o))
        return "BoundaryInvertedPendulumSwingUp", s, a, f, 0.02
    if body == "dpend":
        s = np.column_stack([rng.uniform(-3.1, 3.1, n), rng.uniform(-3.5, 3.5, (n, 2)), rng.normal(0, 2, (n, 3))])
        a = rng.uniform(-1.2, 1.2, (n, 1))
        f = lambda st, ac, fr, dt, o: O.dpend_step("boundary_swingup", st, ac[:, 0], fr, dt, o)
        return "BoundaryInvertedDoublePendulumSwingUp", s, a, f, 0.02
    if body == "cheetah":
        q = rng.normal(0, 0.15, (n, 9))
        q[:, 1] = rng.uniform(-0.35, 0.3, n)
        q[: n // 4, 3:] = rng.uniform(-1.3, 1.3, (n // 4, 6))
        s = np.concatenate([q, rng.normal(0, 1.5, (n, 9))], axis=1)
        a = rng.uniform(-1.3, 1.3, (n, 6))
        f = lambda st, ac, fr, dt, o: (lambda r: (r[0], r[0], r[1], r[2]))(O.cheetah_step(st, ac, fr, dt, o))
        return "HalfCheetahRunning", s, a, f, 0.002
    q = rng.normal(0, 0.1, (n, 6))
    q[:, 1] = 1.25 + rng.uniform(-0.08, 0.3, n)
    q[:, 3:5] = -np.abs(rng.normal(0, 0.3, (n, 2)))
    s = np.concatenate([q, rng.normal(0, 1.5, (n, 6))], axis=1)
    a = rng.uniform(-1.3, 1.3, (n, 3))
    f = lambda st, ac, fr, dt, o: (lambda r: (r[0], r[0], r[1], r[2]))(O.hopper_step(st, ac, fr, dt, o))
    return "HopperRunning", s, a, f, 0.002


@pytest.mark.parametrize("integrator", INTEGRATORS)
@pytest.mark.parametrize("body", ["ip", "dpend", "cheetah", "hopper"])
def test_integrators_onestep_vs_oracle(body, integrator):
    from oracle import oracle as O

    rng = np.random.default_rng(11)
    n = 777
    name, s0, act, ostep, dt = _case(body, rng, n)
    act32 = act.astype(np.float32)
    for fr in (1, 3):
        eng = _engine(name, n, freq_rate=fr, real_time_scale=dt, integrator=integrator)
        eng.set_state(s0)
        obs, rew, done = eng.step(torch.as_tensor(act32, device=eng.device))
        o_st, o_obs, o_rew, o_term = ostep(s0, act32.astype(np.float64), fr, dt, O.opts(integrator))
        assert rel_err(eng.get_state().cpu().numpy(), o_st, floor=1.0) <= 1e-9, (body, integrator, fr)
        assert rel_err(obs.cpu().numpy(), o_obs) <= 1e-5
        assert rel_err(rew.cpu().numpy(), o_rew) <= 2e-5
        assert np.array_equal((done.cpu().numpy() & 1).astype(bool), o_term)


def test_ip_euler_body_path_equals_staged_path():
    """The InvertedPendulum has two rollout kernels: the staged 4-state one (euler, no noise) and the generic
    Body one (everything else).  With a vanishing noise sigma the Body path runs the same Euler arithmetic."""
    rng = np.random.default_rng(2)
    n, T = 640, 40
    s0 = np.column_stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-3, 3, n), rng.normal(0, 1, n), rng.normal(0, 2, n)])
    acts = torch.as_tensor(rng.uniform(-3, 3, (T, n, 1)).astype(np.float32), device="cuda")
    name = "BoundaryInvertedPendulumSwingUp"
    a = _engine(name, n, freq_rate=2)
    b = _engine(name, n, freq_rate=2, obs_noise=[1e-30, 0, 0, 0])
    a.set_state(s0)
    b.set_state(s0)
    oa, ra, da = a.rollout(acts)
    ob, rb, db = b.rollout(acts)
    assert rel_err(ob[:5].cpu().numpy(), oa[:5].cpu().numpy()) <= 1e-6 and torch.equal(da[:5], db[:5])
    assert rel_err(rb[:5].cpu().numpy(), ra[:5].cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("layout", ["iid", "shared"])
@pytest.mark.parametrize("body", ["ip", "dpend", "cheetah", "hopper"])
def test_obs_noise_draw_for_draw_vs_oracle(body, layout):
    """Per-substep observation noise (mujoco_env.py:98-104): same counter-based draws on both sides."""
    from oracle import oracle as O

    rng = np.random.default_rng(13)
    n, fr = 300, 2
    name, s0, act, ostep, dt = _case(body, rng, n)
    act32 = act.astype(np.float32)
    sig = (0.01, 0.03)
    eng = _engine(name, n, freq_rate=fr, real_time_scale=dt, integrator="rk4", obs_noise=sig, noise_layout=layout, seed=21,
                  env_index_offset=1000)
    eng.set_state(s0)
    st = s0
    for t in range(3):  # the step index is a counter word: three consecutive steps use three different draws
        obs, rew, done = eng.step(torch.as_tensor(act32, device=eng.device))
        st, o_obs, o_rew, _ = ostep(st, act32.astype(np.float64), fr, dt,
                                    O.opts("rk4", obs_noise=sig, shared=layout == "shared", seed=21, env_offset=1000, step_index=t))
        assert rel_err(eng.get_state().cpu().numpy(), st, floor=1.0) <= 1e-6, (body, layout, t)  # float32 Box-Muller: exact value (oracle) vs hardware transcendentals (device)
        st = eng.get_state().cpu().numpy()  # re-synchronise
    clean = _engine(name, n, freq_rate=fr, real_time_scale=dt, integrator="rk4")
    clean.set_state(s0)
    clean.step(torch.as_tensor(act32, device=eng.device))
    eng.set_state(s0, reset_counters=True)
    eng.step(torch.as_tensor(act32, device=eng.device))
    d = (eng.get_state() - clean.get_state()).cpu().numpy()
    assert 0.002 < np.abs(d).mean() < 0.2  # noise of the last substep (~sigma) plus the propagated first one
    if layout == "shared":  # one substep, then only the draw separates the two states: identical over qpos / over qvel
        one = _engine(name, n, freq_rate=1, real_time_scale=dt, integrator="rk4", obs_noise=sig, noise_layout=layout, seed=21)
        ref = _engine(name, n, freq_rate=1, real_time_scale=dt, integrator="rk4")
        for e in (one, ref):
            e.set_state(s0)
            e.step(torch.as_tensor(act32, device=eng.device))
        d1 = (one.get_state() - ref.get_state()).cpu().numpy()
        half = d1.shape[1] // 2
        assert np.allclose(d1[:, :half], d1[:, :1], atol=1e-9) and np.allclose(d1[:, half:], d1[:, half : half + 1], atol=1e-9)
        assert d1[:, 0].std() == pytest.approx(0.01, rel=0.2) and d1[:, half].std() == pytest.approx(0.03, rel=0.2)


def test_noise_statistics_and_tuple_sigmas():
    n = 8192
    eng = _engine("HopperRunning", n, freq_rate=1, real_time_scale=1e-9, integrator="euler", obs_noise=(0.02, 0.05), seed=9)
    s0 = np.tile([0, 3.0, 0, -0.2, -0.2, 0, 0, 0, 0, 0, 0, 0.0], (n, 1))
    eng.set_state(s0)
    eng.step(torch.zeros((n, 3), device=eng.device))
    d = eng.get_state().cpu().numpy() - s0
    assert d[:, :6].std() == pytest.approx(0.02, rel=0.03) and d[:, 6:].std() == pytest.approx(0.05, rel=0.03)
    assert abs(d.mean()) < 2e-3 and abs(np.corrcoef(d[:, 0], d[:, 1])[0, 1]) < 0.05
    per = [0.0] * 12
    per[2], per[8] = 0.1, 0.2  # the dict form {2: (0.1, 0.2)} reduced to per-coordinate sigmas
    eng = _engine("HopperRunning", n, freq_rate=1, real_time_scale=1e-9, integrator="euler", obs_noise=per, seed=9)
    eng.set_state(s0)
    eng.step(torch.zeros((n, 3), device=eng.device))
    d = eng.get_state().cpu().numpy() - s0
    assert d[:, 2].std() == pytest.approx(0.1, rel=0.03) and d[:, 8].std() == pytest.approx(0.2, rel=0.03)
    assert np.abs(np.delete(d, [2, 8], axis=1)).max() < 1e-7


def test_init_layouts_on_device():
    from oracle import oracle as O

    for name, nv in (("BoundaryInvertedPendulumBalancing", 2), ("HalfCheetahRunning", 9), ("HopperRunning", 6)):
        for layout in ("iid", "shared"):
            eng = _engine(name, 512, init_noise=(0.1, 0.2), noise_layout=layout, seed=4, env_index_offset=50)
            eng.reset(4)
            s = eng.get_state().cpu().numpy()
            base = np.zeros(2 * nv)
            if name == "HopperRunning":
                base[1] = 1.25
            for e in (0, 5, 511):
                want = O.body_init(4, 50 + e, 0, nv, 0.1, 0.2, shared=layout == "shared") + base
                # exact Box-Muller (oracle) vs the hardware float32 transcendentals: |dz| <= ~1.2e-6 (tools/bm_accuracy.hip)
                assert np.abs(s[e] - want).max() <= 0.2 * 1.5e-6, (name, layout, e)
            if layout == "shared":
                assert np.allclose(s[:, :nv] - base[:nv], (s[:, :1] - base[0]), atol=1e-12)


def test_abi_v1_config_still_accepted():
    """A caller compiled against ABI version 1 passes the 64-byte emei_config: Euler, i.i.d. init noise."""
    import ctypes as C

    from emei_amd import _lib as L

    class CfgV1(C.Structure):
        _fields_ = L.EmeiConfig._fields_[:11]

    assert C.sizeof(CfgV1) == L.CONFIG_SIZE_V1
    cfg = CfgV1(C.sizeof(CfgV1), L.ENV_IDS["HalfCheetahRunning"], 256, 4, 0, 0.002, 0, torch.cuda.current_device(), 3, 0, 0.1)
    h = C.c_void_p()
    create = L.lib().emei_create
    create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    try:
        L.check(create(C.byref(cfg), C.byref(h)))
    finally:
        create.argtypes = L.SYMBOLS["emei_create"][1]
    L.check(L.lib().emei_reset(h, 3, None))
    out = torch.empty((256, 18), dtype=torch.float64, device="cuda")
    L.check(L.lib().emei_get_state(h, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert float(out.std()) == pytest.approx(0.1, rel=0.05)
    L.lib().emei_destroy(h)


@pytest.mark.parametrize("integrator", INTEGRATORS)
@pytest.mark.parametrize("body", ["ip", "cheetah", "hopper"])
def test_batch_next_obs_equals_a_step_from_that_state(body, integrator):
    """get_batch_next_obs (core.py:190-193, abstract in the reference): one env-step from caller-supplied
    float32 observations == set_state(those rows) + step, for every body whose observation determines the state."""
    from emei_amd import engine as E

    rng = np.random.default_rng(17)
    n = 333
    name, s0, act, _, dt = _case(body, rng, n)
    if body == "ip":
        s0[:, 1] = (s0[:, 1] + np.pi) % (2 * np.pi) - np.pi  # observations carry the wrapped angle
    obs32 = torch.as_tensor(s0.astype(np.float32), device="cuda")
    act32 = torch.as_tensor(act.astype(np.float32), device="cuda")
    nxt = E.batch_next_obs(name, obs32, act32 if body != "ip" else act32.reshape(-1), dt, 2, "ref", integrator)
    eng = _engine(name, n, freq_rate=2, real_time_scale=dt, integrator=integrator)
    eng.set_state(obs32.double().cpu().numpy())
    obs, _, _ = eng.step(act32)
    assert rel_err(nxt.cpu().numpy(), obs.cpu().numpy()) <= 1e-6, (body, integrator)


def test_batch_next_obs_env_surface_and_unsupported():
    import emei_amd
    from emei_amd import engine as E

    env = emei_amd.HopperRunningEnv()
    o = np.tile([0, 1.25, 0, -0.1, -0.2, 0.05, 0, 0, 0, 0, 0, 0.0], (8, 1))
    a = np.zeros((8, 3))
    with pytest.raises(AssertionError):  # `assert self.frozen`, core.py:191
        env.get_batch_next_obs(o, action=a)
    np.random.seed(0)
    env.reset()
    env.freeze()
    nxt = env.get_batch_next_obs(o, action=a)
    env.unfreeze()
    assert nxt.shape == (8, 12) and nxt.dtype == np.float64 and np.all(nxt[:, 1] < 1.25) and np.allclose(nxt, nxt[:1])
    with pytest.raises(NotImplementedError):  # the double pendulum's observation wrap is not invertible
        E.batch_next_obs("BoundaryInvertedDoublePendulumSwingUp", torch.zeros((4, 6), device="cuda"), torch.zeros((4, 1), device="cuda"))


@pytest.mark.parametrize("integrator", INTEGRATORS)
@pytest.mark.parametrize("body", ["ip", "dpend", "cheetah", "hopper"])
def test_integrators_f32_mode_tracks_oracle(body, integrator):
    """Every (body, integrator) instantiation of the float32 kernels against the float64 oracle, on benign states
    (no deep contact / far-beyond-limit rows, whose stiff terms amplify float32 rounding): a gross check that each
    template instantiation computes the same physics."""
    from oracle import oracle as O

    rng = np.random.default_rng(23)
    n = 320
    name, s0, act, ostep, dt = _case(body, rng, n)
    if body in ("cheetah", "hopper"):
        nv = s0.shape[1] // 2
        s0[:, 1] += 1.0  # in flight: no contacts
        s0[:, 3:nv] = np.clip(s0[:, 3:nv], -0.3, 0.3) if body == "cheetah" else -np.abs(np.clip(s0[:, 3:nv], -0.3, 0.3)) * [1, 1, 0.5]
    else:
        s0[:, 0] = np.clip(s0[:, 0], -1.5, 1.5)
    act32 = act.astype(np.float32)
    eng = _engine(name, n, freq_rate=2, real_time_scale=dt, integrator=integrator, precision="f32")
    eng.set_state(s0)
    obs, rew, done = eng.step(torch.as_tensor(act32, device=eng.device))
    o_st, o_obs, o_rew, o_term = ostep(s0.astype(np.float32).astype(np.float64), act32.astype(np.float64), 2, dt, O.opts(integrator))
    assert rel_err(obs.cpu().numpy(), o_obs) <= 2e-3, (body, integrator)
