"""InvertedPendulum kernels against the oracle (float64 restatement of the same MuJoCo 2-DoF model +
emei's forward-Euler override) and against the golden reward/terminal vectors of the reference.
Parity with libmujoco itself is unpinned (no MuJoCo in the image): see DESIGN.md."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RTOL = 1e-5
VARIANTS = {
    "rebound_balancing": "ReboundInvertedPendulumBalancing",
    "boundary_balancing": "BoundaryInvertedPendulumBalancing",
    "rebound_swingup": "ReboundInvertedPendulumSwingUp",
    "boundary_swingup": "BoundaryInvertedPendulumSwingUp",
}


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


def _states(rng, n):
    s = np.empty((n, 4))
    s[:, 0] = rng.uniform(-2.3, 2.3, n)  # beyond the +-2 rail on purpose: limit force engaged
    s[:, 1] = rng.uniform(-12, 12, n)
    s[:, 2] = rng.normal(0, 2, n)
    s[:, 3] = rng.normal(0, 5, n)
    s[: n // 4] = rng.standard_normal((n // 4, 4)) * 5e-3
    return s


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("fr,dt", [(1, 0.02), (4, 0.02), (4, 0.002)])
@pytest.mark.parametrize("precision", ["ref", "f32"])
def test_onestep_vs_oracle(variant, fr, dt, precision):
    from oracle import oracle as O

    rng = np.random.default_rng(42)
    n = 2048
    s0 = _states(rng, n)
    act = rng.uniform(-3.5, 3.5, n).astype(np.float32)  # beyond ctrlrange: clipping exercised
    eng = _engine(VARIANTS[variant], n, freq_rate=fr, real_time_scale=dt, precision=precision)
    eng.set_state(s0)
    obs, rew, done = eng.step(torch.as_tensor(act, device=eng.device))
    o_st, o_obs, o_rew, o_term = O.ip_step(variant, s0, act.astype(np.float64), fr, dt)
    # float32 mode: O(1) float32 intermediates (stiff rail force x dt) set an ABSOLUTE error, stated as such (floor 1.0)
    tol, fl = (RTOL, 1e-3) if precision == "ref" else (2e-4, 1.0)
    assert rel_err(obs.cpu().numpy(), o_obs, floor=fl) <= tol
    assert rel_err(rew.cpu().numpy(), o_rew, floor=fl) <= tol
    if precision == "ref":
        assert rel_err(eng.get_state().cpu().numpy(), o_st, floor=1e-30) <= 1e-9
        assert rel_err(eng.get_obs().cpu().numpy(), o_obs, floor=1e-30) <= 1e-9
        # masks bit-exact away from the thresholds (cos theta vs 0.9 / 0, x vs +-2)
        y = np.cos(o_obs[:, 1])
        clear = (np.abs(y - 0.9) > 1e-9) & (np.abs(y) > 1e-9) & (np.abs(np.abs(o_obs[:, 0]) - 2) > 1e-9)
        assert np.array_equal((done.cpu().numpy() & 1)[clear].astype(bool), o_term[clear])


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_trajectory_vs_oracle(variant):
    """Open loop in segments: the float64 MuJoCo-style dynamics (stiff rail constraint, explicit Euler,
    no float32 rounding of the derivative to absorb last-bit differences) amplify 1e-16 by many orders
    over hundreds of substeps, so the oracle is re-synchronised to the device state every 25 steps
    (100 substeps) and each segment must agree to 1e-5."""
    from oracle import oracle as O

    rng = np.random.default_rng(7)
    n, T, fr, seg = 64, 200, 4, 25
    s0 = rng.standard_normal((n, 4)) * 5e-3
    acts = rng.uniform(-3, 3, (T, n)).astype(np.float32)
    eng = _engine(VARIANTS[variant], n, freq_rate=fr, precision="ref")
    eng.set_state(s0)
    for t0 in range(0, T, seg):
        st = eng.get_state().cpu().numpy()
        obs, rew, done = eng.rollout(torch.as_tensor(acts[t0 : t0 + seg], device=eng.device))
        o_obs = np.empty((seg, n, 4))
        o_rew = np.empty((seg, n))
        for t in range(seg):
            st, o_obs[t], o_rew[t], _ = O.ip_step(variant, st, acts[t0 + t].astype(np.float64), fr, 0.02)
        got = obs.cpu().numpy().astype(np.float64)
        dth = np.angle(np.exp(1j * (got[..., 1] - o_obs[..., 1])))  # theta wraps at +-pi: compare on the circle
        got[..., 1] = o_obs[..., 1] + dth
        assert rel_err(got, o_obs) <= RTOL, t0
        assert rel_err(rew.cpu().numpy(), o_rew) <= RTOL, t0


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_reward_terminal_vs_golden(mujoco_golden, variant):
    from emei_amd import engine as E

    g = mujoco_golden
    obs = g["ip_obs"]
    o64 = torch.as_tensor(obs, dtype=torch.float64, device="cuda")  # float64 rows, as the reference computes (inverted_pendulum.py:73-183)
    rew = E.batch_reward(VARIANTS[variant], o64).cpu().numpy()
    term = E.batch_terminal(VARIANTS[variant], o64).cpu().numpy()
    assert rel_err(rew, g[f"ip_{variant}_reward"][:, 0]) <= RTOL
    assert np.array_equal(term, g[f"ip_{variant}_terminal"][:, 0])  # every golden row, bit for bit


def test_behaviour_like_reference_tests():
    """test/test_envs/test_mujoco/test_inverted_pendulum.py:14-62 of the reference, vectorised:
    under random actions Rebound/Boundary Balancing and Boundary SwingUp terminate eventually,
    Rebound SwingUp must NOT terminate within 100 steps."""
    n, T = 256, 1000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    acts = (torch.rand((T, n), device="cuda", generator=gen) * 6 - 3).float()
    for variant, must in (("rebound_balancing", True), ("boundary_balancing", True), ("boundary_swingup", True)):
        eng = _engine(VARIANTS[variant], n, freq_rate=1, init_noise=5e-3, seed=1)
        eng.reset(1)
        _, _, done = eng.rollout(acts)
        # "terminates eventually" (the reference loops one env until terminal): nearly every env within the horizon
        assert (float(((done & 1) != 0).any(dim=0).float().mean()) > 0.97) == must, variant
    eng = _engine(VARIANTS["rebound_swingup"], n, freq_rate=1, init_noise=5e-3, seed=1)
    eng.reset(1)
    _, _, done = eng.rollout(acts[:100].contiguous())
    assert not bool((done & 1).any())
    # the rail holds the cart: |x| stays close to the +-2 range under constant full push
    eng = _engine(VARIANTS["rebound_swingup"], n, freq_rate=1, init_noise=5e-3, seed=1)
    eng.reset(1)
    obs, _, _ = eng.rollout(torch.full((400, n), 3.0, device="cuda"))
    assert float(obs[..., 0].abs().max()) < 2.5 and float(obs[-1, :, 0].min()) > 1.5


def test_device_reset_matches_oracle_spec():
    from oracle import oracle as O

    eng = _engine(VARIANTS["boundary_swingup"], 512, init_noise=5e-3, seed=21, env_index_offset=77)
    eng.reset(21)
    got = eng.get_state().cpu().numpy()
    want = np.stack([O.ip_init_f32(21, 77 + i, 0, 5e-3) for i in range(512)]).astype(np.float64)
    # the oracle holds the exact Box-Muller value; the device's hardware sin / cos / log2 / sqrt are within 1.3e-7 and 4.9e-7
    # of it (tools/bm_accuracy.hip, all 2^24 inputs): |dz| <= 4 * 1.3e-7 + 4.9e-7 ~ 1e-6 at radius 4, times sigma = 5e-3
    assert np.abs(got - want).max() <= 1e-8


@pytest.mark.parametrize("variant", ["boundary_swingup", "boundary_balancing"])
def test_autoreset_rollout_vs_oracle(variant):
    """Device-side resets of the staged kernel (spare initial states, the per-lane spare flag, TimeLimit) against the oracle
    stepped and reset the same way, over several episodes per env: done codes bit for bit (which step ends an episode, and
    why), observations / rewards to float32 tolerance.  The twin restarts an env from the oracle's exact Box-Muller draw of
    (seed, env, episode); the device's hardware transcendentals differ from it by <= 1e-8 (test_device_reset_matches_oracle_spec),
    which a 30-step episode amplifies to ~1e-6."""
    from oracle import oracle as O

    N, T, seed, off, max_steps, fr, sigma = 256, 96, 9, 4000, 30, 4, 5e-3
    eng = _engine(VARIANTS[variant], N, freq_rate=fr, init_noise=sigma, max_episode_steps=max_steps, seed=seed, env_index_offset=off)
    eng.reset(seed)
    st = eng.get_state().cpu().numpy()
    acts = np.random.default_rng(13).uniform(-3, 3, (T, N)).astype(np.float32)  # +-300 N: the cart leaves the rail within ~20 steps
    obs, rew, done = eng.rollout(torch.as_tensor(acts, device=eng.device), auto_reset=True)
    from emei_amd import _lib as L

    assert eng.last_kernel() == L.KERNEL_PEND_STAGED
    steps, epi = np.zeros(N, np.int64), np.zeros(N, np.int64)
    o_obs, o_rew, o_done = np.empty((T, N, 4)), np.empty((T, N)), np.zeros((T, N), np.uint8)
    for t in range(T):
        st, o_obs[t], o_rew[t], term = O.ip_step(variant, st, acts[t].astype(np.float64), fr, 0.02)
        steps += 1
        o_done[t] = term.astype(np.uint8) | ((steps >= max_steps).astype(np.uint8) << 1)
        for i in np.nonzero(o_done[t])[0]:
            epi[i] += 1
            steps[i] = 0
            st[i] = O.ip_init_f32(seed, off + i, epi[i], sigma).astype(np.float64)
    assert np.array_equal(done.cpu().numpy(), o_done)
    # both kinds of ending (a balancing pole under these pushes never lasts 30 steps); every env restarted at least twice
    assert (o_done & 1).any() and ((o_done & 2).any() or variant == "boundary_balancing") and epi.min() >= 2
    # an ABSOLUTE error set by O(1)-O(10) dynamics (floor 1.0): positions and angles pass through zero
    assert rel_err(obs.cpu().numpy(), o_obs, floor=1.0) <= 2e-5
    assert rel_err(rew.cpu().numpy(), o_rew, floor=1.0) <= 2e-5
    _, e = eng.get_counters()
    assert np.array_equal(e.cpu().numpy(), epi)


def test_full_size_config3_properties():
    """BASELINE configs[2]: 262 144 envs, freq_ratio 4 — rollout == chunked rollouts; obs angle in [-pi, pi)."""
    N, T = 262144, 32
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    acts = (torch.rand((T, N), device="cuda", generator=gen) * 6 - 3).float()
    a = _engine(VARIANTS["boundary_swingup"], N, freq_rate=4, init_noise=5e-3, max_episode_steps=1000, seed=2)
    a.reset(2)
    obs, rew, done = a.rollout(acts, auto_reset=True)
    b = _engine(VARIANTS["boundary_swingup"], N, freq_rate=4, init_noise=5e-3, max_episode_steps=1000, seed=2)
    b.reset(2)
    p = [b.rollout(acts[k : k + 16].contiguous(), auto_reset=True) for k in (0, 16)]
    assert torch.equal(torch.cat([p[0][0], p[1][0]]), obs) and torch.equal(torch.cat([p[0][2], p[1][2]]), done)
    th = obs[..., 1]
    assert float(th.min()) >= -np.pi - 1e-6 and float(th.max()) < np.pi + 1e-6
    # (1 - cos theta) / 2 with the kernels' cosine, which may exceed 1 by one float64 ulp (the reference's libm never does):
    # the reward stays within one ulp of [0, 1]
    assert float(rew.min()) >= -2.3e-16 and float(rew.max()) <= 1 + 2.3e-16 and bool(torch.isfinite(obs).all())


def test_huge_and_nonfinite_angles():
    """Angles beyond the table path's range (|theta| > 1e6 rad) go through the exact modulo-2-pi reduction
    (emei_device.h:trig_reduce_large, v_trig_preop_f64); non-finite angles give NaN.  The oracle's libm is exact for any
    argument, so the velocities of one substep (which carry sin / cos of the OLD angle) pin the reduction."""
    from oracle import oracle as O

    rng = np.random.default_rng(7)
    mags = np.array([1.0000001e6, 3.3e6, 1e7, 7.7e8, 1e10, 2.5e12, 1e15, 4e18, 1e30, 1e100, 8.8e200, 1.7e308])
    theta = np.concatenate([mags, -mags, mags * rng.uniform(0.5, 1.0, len(mags)), rng.uniform(-1e9, 1e9, 476)])
    n = len(theta)
    s0 = np.column_stack([rng.uniform(-1.5, 1.5, n), theta, rng.normal(0, 1, n), np.zeros(n)])  # omega 0: theta stays put
    act = rng.uniform(-3, 3, n).astype(np.float32)
    # (the SwingUp variants: their hinge is free; a Balancing variant at |theta| = 1e6 rad is 1e6 rad beyond its +-90 degree stop)
    for name, variant in (("BoundaryInvertedPendulumSwingUp", "boundary_swingup"), ("ReboundInvertedPendulumSwingUp", "rebound_swingup")):
        eng = _engine(name, n, freq_rate=1, real_time_scale=0.02)
        eng.set_state(s0)
        obs, rew, done = eng.step(torch.as_tensor(act, device=eng.device))
        o_st, o_obs, o_rew, o_term = O.ip_step(variant, s0, act.astype(np.float64), 1, 0.02)
        st = eng.get_state().cpu().numpy()
        # v, omega carry sin / cos of the huge angle.  The oracle adds the offsets as (theta + phi0) + pi, the kernel as
        # theta + (phi0 + pi): the two sums may differ by one ulp of theta, i.e. by dt * |da/dphi| * ulp(theta) <= ulp(theta)
        err = np.abs(st[:, [0, 2, 3]] - o_st[:, [0, 2, 3]]).max(axis=1)
        assert np.all(err <= 1e-9 + 2 * np.spacing(np.abs(theta))), float((err - 2 * np.spacing(np.abs(theta))).max())
    # CartPole (pinned to the reference): the same reduction behind math.sin / math.cos of the reference
    sc = np.column_stack([rng.uniform(-1, 1, n), rng.normal(0, 1, n), theta, np.zeros(n)])
    a = rng.integers(2, size=n)
    eng = _engine("CartPoleSwingUp", n)
    eng.set_state(sc)
    obs, rew, done = eng.step(torch.as_tensor(a, device=eng.device).to(torch.uint8))
    nxt, orew, oterm = O.cartpole_step("swingup", sc, a)
    # the derivative is rounded to float32 before it is accumulated (cartpole.py:60): a last-bit difference of sin / cos
    # may flip that rounding (6e-8 relative of an acceleration of O(10), times dt)
    assert rel_err(eng.get_state().cpu().numpy()[:, [0, 1, 3]], nxt[:, [0, 1, 3]], floor=1.0) <= 5e-8
    assert rel_err(rew.cpu().numpy(), orew) <= 1e-6
    # non-finite angles: NaN out, terminal (inverted_pendulum.py:179-183: not finite -> done)
    bad = np.array([[0.0, np.inf, 0.0, 0.0], [0.0, -np.inf, 0.0, 0.0], [0.0, np.nan, 0.0, 0.0]])
    eng = _engine("BoundaryInvertedPendulumSwingUp", 3)
    eng.set_state(bad)
    obs, rew, done = eng.step(torch.zeros(3, dtype=torch.float32, device=eng.device))
    assert bool(torch.isnan(obs[:, 2:]).all()) and bool((done & 1).all())


@pytest.mark.parametrize("variant", ["rebound_balancing", "boundary_balancing"])
@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_hinge_stop_of_the_balancing_variants(variant, integrator):
    """inverted_pendulum.xml:17: the hinge of the Balancing variants is limited to +-90 degrees (SwingUp frees it).  Only a
    post-terminal state reaches the stop — the reference keeps stepping after `terminal` (mujoco_env.py:157-167 has no reset) —
    so: fall from 1.2 rad without auto-reset, through the staged kernel (euler: T = 48 >= one tile) and the Body kernel (rk4),
    against the oracle re-synchronised every 16 steps; states with BOTH rows (rail and stop) in the one-step sweep."""
    from oracle import oracle as O

    rng = np.random.default_rng(21)
    n = 256
    s0 = np.column_stack([rng.uniform(-1.5, 1.5, n), rng.choice([-1.0, 1.0], n) * rng.uniform(1.0, 1.5, n), rng.normal(0, 1, n), rng.normal(0, 1, n)])
    acts = rng.uniform(-3, 3, (96, n)).astype(np.float32)
    eng = _engine(VARIANTS[variant], n, freq_rate=2, real_time_scale=0.02, integrator=integrator)
    eng.set_state(s0)
    dev = torch.as_tensor(acts, device=eng.device)
    peak = 0.0
    for t0 in range(0, 96, 16):
        st = eng.get_state().cpu().numpy()
        obs, rew, done = eng.rollout(dev[t0:t0 + 16].contiguous())
        for t in range(16):
            st, o_obs, o_rew, o_term = O.ip_step(variant, st, acts[t0 + t].astype(np.float64), 2, 0.02, O.opts(integrator))
            got = obs[t].cpu().numpy().astype(np.float64)
            got[:, 1] = o_obs[:, 1] + np.angle(np.exp(1j * (got[:, 1] - o_obs[:, 1])))
            assert rel_err(got, o_obs, floor=1.0) <= RTOL, (t0, t)
        end = eng.get_state().cpu().numpy()
        assert rel_err(end, st, floor=1.0) <= 1e-7, t0
        peak = max(peak, np.abs(end[:, 1]).max())
    assert np.pi / 2 < peak < np.pi / 2 + 0.2  # the poles reached their stops and were held there
    # one step from states beyond BOTH limits at once
    s1 = np.column_stack([rng.choice([-1.0, 1.0], n) * rng.uniform(1.95, 2.1, n), rng.choice([-1.0, 1.0], n) * rng.uniform(1.5, 1.7, n),
                          rng.normal(0, 2, n), rng.normal(0, 3, n)])
    a1 = rng.uniform(-3, 3, n).astype(np.float32)
    eng.set_state(s1)
    eng.step(torch.as_tensor(a1, device=eng.device))
    want = O.ip_step(variant, s1, a1.astype(np.float64), 2, 0.02, O.opts(integrator))[0]
    assert rel_err(eng.get_state().cpu().numpy(), want, floor=1.0) <= 1e-9


@pytest.mark.parametrize("variant", ["boundary_balancing", "rebound_balancing"])
def test_a_lane_at_the_hinge_stop_does_not_change_its_wave_mates(variant):
    """The staged kernel takes a wave-uniform cold branch when some lane is at the hinge stop; the other lanes must leave it
    with the bits of the hot path — also those beyond the rail, whose one-row force has a closed form there (pendulum_envs.h:
    InvPend::substep).  Same wave twice: lane 5 hanging beyond its stop, or upright; every other lane bit for bit."""
    rng = np.random.default_rng(33)
    n, T = 64, 48
    base = np.column_stack([rng.uniform(-2.2, 2.2, n), rng.uniform(-0.2, 0.2, n), rng.normal(0, 2, n), rng.normal(0, 1, n)])
    assert (np.abs(base[:, 0]) > 2.0).sum() >= 3  # several lanes start beyond the rail (post-terminal for Boundary, rebounding otherwise)
    acts = torch.as_tensor(rng.uniform(-3, 3, (T, n)).astype(np.float32), device="cuda:0")
    out = []
    for theta5 in (0.05, 1.65):
        s0 = base.copy()
        s0[5, 1] = theta5
        eng = _engine(VARIANTS[variant], n, freq_rate=2, real_time_scale=0.02)
        eng.set_state(s0)
        obs, rew, done = eng.rollout(acts)
        out.append((obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), eng.get_state().cpu().numpy()))
    others = np.arange(n) != 5
    assert np.abs(out[1][3][5, 1]) > np.pi / 2 - 0.05  # lane 5 really sat at its stop
    for a, b in zip(out[0], out[1]):
        a, b = (a[:, others], b[:, others]) if a.shape[0] == T else (a[others], b[others])
        assert np.array_equal(a, b)


def test_angle_wrap_at_its_boundaries():
    """inverted_pendulum.py:45-49: theta -> (theta + pi) % (2 pi) - pi.  The kernels test the wrapped result once (|o| >= pi) and
    repair in a cold branch (emei_device.h:wrap_pi); these angles sit on and next to the wrap points, where that branch runs:
    bit for bit against NumPy's floored modulo, through emei_step (generic kernel), a staged rollout and emei_get_obs."""
    pi = np.pi
    base = np.array([0.0, pi, -pi, 3 * pi, -3 * pi, 2 * pi, -2 * pi, 101 * pi, -101 * pi, 600.0, -600.0, 1e5 * pi, -1e5 * pi])
    theta = np.concatenate([base, np.nextafter(base, np.inf), np.nextafter(base, -np.inf), base + 1e-9, base - 1e-9,
                            np.random.default_rng(3).uniform(-50, 50, 999)])
    n = 1024 * ((len(theta) + 1023) // 1024)
    theta = np.resize(theta, n)
    want = (theta + pi) % (2 * pi) - pi

    def same(got, ref):
        # bit for bit — except on the wrap point itself, where NumPy's own remainder can round up to the modulus (theta one ulp
        # below -pi gives +pi there, -pi here: the same angle, one representative each)
        at_wrap = (np.abs(np.abs(got.astype(np.float64)) - pi) < 1e-6) & (np.abs(np.abs(ref.astype(np.float64)) - pi) < 1e-6)
        return bool(np.all((got == ref) | at_wrap))

    s0 = np.zeros((n, 4))
    s0[:, 1] = theta  # at rest: the position update uses the OLD velocity (mujoco_env.py:189-191), so theta stays put
    zero = torch.zeros((16, n), dtype=torch.float32, device="cuda")
    for name in ("BoundaryInvertedPendulumSwingUp", "ReboundInvertedPendulumBalancing"):
        eng = _engine(name, n, freq_rate=1, real_time_scale=0.02)
        eng.set_state(s0)
        assert same(eng.get_obs().cpu().numpy()[:, 1], want)
        obs, _, _ = eng.step(zero[0])
        assert np.array_equal(eng.get_state().cpu().numpy()[:, 1], theta)
        assert same(obs.cpu().numpy()[:, 1], want.astype(np.float32))
        eng.set_state(s0)
        obs, _, _ = eng.rollout(zero[:8].contiguous())  # the staged kernel; the first step's angle is still theta
        assert same(obs[0].cpu().numpy()[:, 1], want.astype(np.float32))
        eng.close()
