"""The CPU oracle against the golden vectors produced by the real reference (oracle/gen_golden.py).
This is what pins the oracle: bit-exact for the first-party CartPole arithmetic."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle as O


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("name", ["swingup", "balancing"])
@pytest.mark.parametrize("fr,dt", [(1, 0.02), (4, 0.02), (2, 0.01)])
def test_cartpole_onestep_bit_exact(cartpole_golden, name, fr, dt):
    g = cartpole_golden
    tag = f"onestep_{name}_fr{fr}_dt{dt}"
    ok = ~g[tag + "_raised"]  # rows where the reference itself raises (math.cos(inf)) have no answer
    nxt, rew, term = O.cartpole_step(name, g[f"onestep_{name}_state"], g[f"onestep_{name}_action"], fr, dt)
    assert _same(nxt[ok], g[tag + "_next"][ok])
    assert np.array_equal(term[ok], g[tag + "_terminal"][ok])
    assert rel_err(rew[ok], g[tag + "_reward"][ok], floor=1e-300) <= 4e-16  # numpy cos vs libm cos
    # the NumPy restatement agrees with the C one
    n2, r2, t2 = O.cartpole_step_numpy(name, g[f"onestep_{name}_state"], g[f"onestep_{name}_action"], fr, dt)
    assert rel_err(n2[ok], nxt[ok], floor=1e-30) <= 1e-6 and np.array_equal(t2[ok], term[ok])


@pytest.mark.parametrize("name", ["swingup", "balancing"])
@pytest.mark.parametrize("fr", [1, 4])
@pytest.mark.parametrize("seed", range(4))
def test_cartpole_trajectory_bit_exact(cartpole_golden, name, fr, seed):
    """BASELINE configs[0]: 1 env, 1000 CPU steps, here for 16 (env, freq_rate, seed) combinations."""
    g = cartpole_golden
    tag = f"traj_{name}_fr{fr}_seed{seed}"
    st, rew, term = O.cartpole_rollout(name, g[tag + "_states"][:1], g[tag + "_actions"][:, None], fr, 0.02)
    assert _same(st[:, 0], g[tag + "_states"])
    assert np.array_equal(term[:, 0], g[tag + "_terminal"])
    assert rel_err(rew[:, 0], g[tag + "_reward"], floor=1e-300) <= 4e-16


@pytest.mark.parametrize("name", ["swingup", "balancing"])
@pytest.mark.parametrize("fr,dt", [(1, 0.02), (4, 0.02), (2, 0.01)])
def test_cartpole_rk4_onestep_bit_exact(cartpole_rk4_golden, name, fr, dt):
    """The classic-control RK4 branch (base_control.py:165-170), the reference's float32 / float64 promotion chain bit for bit."""
    g = cartpole_rk4_golden
    tag = f"onestep_{name}_fr{fr}_dt{dt}"
    ok = ~g[tag + "_raised"]
    nxt, rew, term = O.cartpole_step(name, g[f"onestep_{name}_state"], g[f"onestep_{name}_action"], fr, dt, method="rk4")
    assert _same(nxt[ok], g[tag + "_next"][ok])
    assert np.array_equal(term[ok], g[tag + "_terminal"][ok])
    assert rel_err(rew[ok], g[tag + "_reward"][ok], floor=1e-300) <= 4e-16


@pytest.mark.parametrize("name", ["swingup", "balancing"])
@pytest.mark.parametrize("fr", [1, 4])
@pytest.mark.parametrize("seed", range(4))
def test_cartpole_rk4_trajectory_bit_exact(cartpole_rk4_golden, name, fr, seed):
    g = cartpole_rk4_golden
    tag = f"traj_{name}_fr{fr}_seed{seed}"
    st, rew, term = O.cartpole_rollout(name, g[tag + "_states"][:1], g[tag + "_actions"][:, None], fr, 0.02, method="rk4")
    assert _same(st[:, 0], g[tag + "_states"])
    assert np.array_equal(term[:, 0], g[tag + "_terminal"])
    assert rel_err(rew[:, 0], g[tag + "_reward"], floor=1e-300) <= 4e-16


def test_cartpole_rk4_differs_from_euler(cartpole_golden, cartpole_rk4_golden):
    """... and is not the Euler branch under another name: same start, same actions, a different trajectory."""
    g = cartpole_rk4_golden
    st_e, _, _ = O.cartpole_rollout("swingup", g["traj_swingup_fr1_seed0_states"][:1], g["traj_swingup_fr1_seed0_actions"][:, None], 1, 0.02)
    assert np.abs(st_e[100, 0] - g["traj_swingup_fr1_seed0_states"][100]).max() > 1e-4


@pytest.mark.parametrize("name", ["swingup", "balancing"])
def test_reference_style_loop_equals_the_c_oracle(name):
    """bench.py's `cpu_baseline.reference_style` (oracle.cartpole_reference_style_loop: one env, a CPython call per step, scalar
    math.sin / cos, float32 np.array derivative — the call pattern of base_control.py:61-83 and of BASELINE configs[0]) computes
    what the pinned C oracle computes, bit for bit, over 1000 steps."""
    st, rew, term = O.cartpole_reference_style_loop(name, 1000, 0)
    s0 = O.cartpole_init_state_host(name, 0, 1)
    acts = np.random.default_rng(1).integers(2, size=1000)
    S, R, T = O.cartpole_rollout(name, s0, acts[:, None])
    assert np.array_equal(S[-1, 0], st) and np.array_equal(T[:, 0], term)
    assert rel_err(rew, R[:, 0], floor=1e-300) <= 4e-16


def test_baseline_md_first_rows(cartpole_golden):
    """The values quoted in BASELINE.md / SURVEY.md 8c: reset(seed=0) then actions 0, 1, 1."""
    s0 = O.cartpole_init_state_host("swingup", 0, 1)
    assert np.allclose(s0[0], [0.01369617, -0.02302133, 3.09569001, -0.04834724], atol=5e-9)
    st, rew, term = O.cartpole_rollout("swingup", s0, np.array([[0], [1], [1]]))
    assert np.allclose(st[1, 0], [0.01323574, -0.21745584, 3.09472306, -0.32620114], atol=5e-9)
    assert np.allclose(st[3, 0], [0.00845284, 0.17415616, 3.08781706, 0.28994059], atol=5e-9)
    assert rew[0, 0] == pytest.approx(0.0005490891410523391, rel=1e-12) and not term.any()


@pytest.mark.parametrize("name", ["swingup", "balancing"])
def test_cartpole_reset_and_batch_functions(cartpole_golden, name):
    g = cartpole_golden
    for seed in range(16):
        assert _same(O.cartpole_init_state_host(name, seed, 1)[0], g[f"reset_{name}_seeds0_15"][seed])
    # drawn after reset(seed=7), which consumes the first row of the stream (base_control.py:44-46)
    assert _same(O.cartpole_init_state_host(name, 7, 9)[1:], g[f"batchinit_{name}_seed7_B8"])
    obs = g[f"batch_{name}_obs"]
    assert np.array_equal(O.cartpole_terminal(name, obs)[:, None], g[f"batch_{name}_terminal"])
    assert rel_err(O.cartpole_reward(name, obs)[:, None], g[f"batch_{name}_reward"], floor=1e-300) <= 4e-16


@pytest.mark.parametrize("variant", ["rebound_balancing", "boundary_balancing", "rebound_swingup", "boundary_swingup"])
def test_invpend_reward_terminal(mujoco_golden, variant):
    g = mujoco_golden
    obs = g["ip_obs"]
    assert np.array_equal(O.ip_terminal(variant, obs)[:, None], g[f"ip_{variant}_terminal"])
    assert rel_err(O.ip_reward(variant, obs)[:, None], g[f"ip_{variant}_reward"], floor=1e-300) <= 4e-16


def test_invpend_wrap_and_euler_rule(mujoco_golden):
    g = mujoco_golden
    assert np.array_equal(O.ip_wrap(g["ip_wrap_in"][:, 1]), g["ip_wrap_out"][:, 1])
    # forward-Euler position rule (mujoco_env.py:189-191) through one oracle substep: q' = q + dt*v_old
    for dt in (0.02, 0.002):
        qp, qv = g[f"euler_ip_dt{dt}_qpos"], g[f"euler_ip_dt{dt}_qvel"]
        st = np.concatenate([qp * 0.1, qv], axis=1)  # keep x inside the rail so no limit force acts
        nxt, _, _, _ = O.ip_step("boundary_swingup", st, np.zeros(len(st)), 1, dt)
        assert rel_err(nxt[:, :2], st[:, :2] + dt * st[:, 2:], floor=1e-30) <= 1e-15


def test_invpend_model_constants():
    """inertia-from-geom of assets/inverted_pendulum.xml: capsule masses with density 1000."""
    m = O.ip_model()
    assert m.mc == pytest.approx(1000 * (np.pi * 0.01 * 0.2 + 4 / 3 * np.pi * 1e-3), rel=1e-14)  # 10.472 kg
    assert m.mp == pytest.approx(5.0186, rel=1e-4)
    assert m.r == pytest.approx(0.3, rel=1e-5) and 0 < m.phi0 < 2e-3
    assert m.gear == 100 and (m.x_lo, m.x_hi) == (-2.0, 2.0) and m.g == 9.81


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (counter = (env lo, env hi, episode, block), key = seed)."""
    assert [int(x) for x in O.philox(0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(x) for x in O.philox(2**64 - 1, 2**64 - 1, 2**32 - 1, 2**32 - 1)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert [int(x) for x in O.philox(0x299F31D0A4093822, 0x85A308D3243F6A88, 0x13198A2E, 0x03707344)] == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_device_reset_distribution():
    s = np.stack([O.cartpole_init_f32("swingup", 3, e, 0) for e in range(4096)])
    s[:, 2] -= np.float32(np.pi)
    assert (np.abs(s) <= 0.05 + 1e-6).all() and abs(s.mean()) < 2e-3 and s.std() == pytest.approx(0.1 / 12**0.5, rel=0.05)
    z = np.stack([O.ip_init_f32(3, e, 0, 1.0) for e in range(4096)])
    assert abs(z.mean()) < 0.05 and z.std() == pytest.approx(1.0, rel=0.05)


def test_gaussian_draws_follow_the_stated_box_muller_spec():
    """oracle/integrators.h: z = sqrt(-2 ln u1) (cos, sin)(2 pi t), u1 = ((a >> 8) + 1) / 2^24, t = (b >> 8) / 2^24 from the
    Philox words (a, b), (c, d) of block 0 — the exact value rounded once to float32.  Restated here in NumPy float64 from
    the Philox known-answer-tested words; the device's hardware transcendentals are held to this value on the GPU
    (test_gpu_invpend.py::test_device_reset_matches_oracle_spec)."""
    for seed, env, epi in ((3, 0, 0), (21, 77, 0), (2**40 + 5, 2**33 + 1, 7)):
        w = [int(x) for x in O.philox(seed, env, epi, 0)]
        want = []
        for a, b in ((w[0], w[1]), (w[2], w[3])):
            u1, t = ((a >> 8) + 1.0) / 2**24, (b >> 8) / 2**24
            rad = np.sqrt(-2.0 * np.log(u1))
            want += [np.float32(rad * np.cos(2 * np.pi * t)), np.float32(rad * np.sin(2 * np.pi * t))]
        got = O.ip_init_f32(seed, env, epi, 1.0)  # sigma = 1: the four normals themselves (x, theta, v, omega)
        assert np.array_equal(np.asarray(got, np.float32), np.asarray(want, np.float32)), (seed, env, epi, got, want)


def test_fused_c_rollout_with_auto_reset_equals_the_per_step_oracle():
    """oracle.cartpole_rollout_autoreset (the CPU twin of the fused device rollout: bench.py's cpu_baseline and the all-env
    GPU parity test) against the per-step oracle with Python-side reset injection (conftest.oracle_autoreset_rollout):
    identical done masks, counters and float64 state; the float32 outputs are the float64 ones rounded."""
    from conftest import oracle_autoreset_rollout
    from oracle import oracle as O

    rng = np.random.default_rng(5)
    for variant, max_steps, fr in (("balancing", 50, 1), ("swingup", 40, 2)):
        n, T = 200, 160
        s0 = O.cartpole_init_state_host(variant, 3, n)
        acts = rng.integers(2, size=(T, n)).astype(np.uint8)
        ids = np.arange(5000, 5000 + n)
        a_obs, a_rew, a_done, a_st = oracle_autoreset_rollout(variant, s0, acts, 11, ids, max_steps, fr)
        b = O.cartpole_rollout_autoreset(variant, s0, acts, 11, ids, max_steps, fr)
        assert np.array_equal(a_done, b["done"]) and np.array_equal(a_st, b["state"])
        assert np.array_equal(a_obs.astype(np.float32), b["obs"]) and np.array_equal(a_rew.astype(np.float32), b["reward"])
        assert int(b["done"].astype(bool).sum()) > n and (b["done"] & 2).any()
        # continuing from the returned counters = one longer call
        c1 = O.cartpole_rollout_autoreset(variant, s0, acts[:70], 11, ids, max_steps, fr)
        c2 = O.cartpole_rollout_autoreset(variant, c1["state"], acts[70:], 11, ids, max_steps, fr, steps=c1["steps"], episode=c1["episode"])
        assert np.array_equal(c2["state"], b["state"]) and np.array_equal(np.concatenate([c1["done"], c2["done"]]), b["done"])


def test_fused_body_rollouts_equal_the_per_step_oracles():
    """oracle.body_rollout (bench.py's cpu_baseline for the MuJoCo-backed bodies: T env-steps per C call, float32 outputs) against
    T calls of ip_step / dpend_step / cheetah_step / hopper_step: the float64 state bit for bit, outputs = the rounded ones."""
    from oracle import oracle as O

    rng = np.random.default_rng(9)
    n, T = 150, 10
    cases = (("ip", "boundary_swingup", 4, 3.0, (), 4, 0.02, "euler", lambda s, a, o: O.ip_step("boundary_swingup", s, a, 4, 0.02, o)),
             ("ip", "rebound_balancing", 4, 3.0, (), 2, 0.02, "rk4", lambda s, a, o: O.ip_step("rebound_balancing", s, a, 2, 0.02, o)),
             ("dp", "boundary_swingup", 6, 1.0, (), 2, 0.02, "rk4", lambda s, a, o: O.dpend_step("boundary_swingup", s, a, 2, 0.02, o)),
             ("cheetah", None, 18, 1.0, (6,), 4, 0.002, "euler", lambda s, a, o: O.cheetah_step(s, a, 4, 0.002, o)),
             ("hopper", None, 12, 1.0, (3,), 4, 0.002, "rk4", lambda s, a, o: O.hopper_step(s, a, 4, 0.002, o)))
    for kind, variant, dim, lo, nu, fr, dt, integ, step in cases:
        s0 = rng.standard_normal((n, dim)) * (0.3 if dim <= 6 else 0.05)
        if kind == "hopper":
            s0[:, 1] += 1.25
        acts = rng.uniform(-lo, lo, (T, n) + nu).astype(np.float32)
        opt = O.opts(integ)
        r = O.body_rollout(kind, variant, s0, acts, fr, dt, opt)
        st = s0.copy()
        for t in range(T):
            out = step(st, acts[t].astype(np.float64), opt)
            st, rew, done = out[0], out[-2], out[-1]
            assert np.array_equal(rew.astype(np.float32), r["reward"][t]), (kind, t)
            assert np.array_equal(np.asarray(done).astype(np.uint8), r["done"][t]), (kind, t)
            if len(out) == 4:  # the pendulums also return the (wrapped) observation
                assert np.array_equal(out[1].astype(np.float32), r["obs"][t]), (kind, t)
            else:
                assert np.array_equal(st.astype(np.float32), r["obs"][t]), (kind, t)
        assert np.array_equal(st, r["state"]), kind
        # the output buffers of a previous call are written again
        r2 = O.body_rollout(kind, variant, s0, acts, fr, dt, opt, reuse=r)
        assert r2["obs"] is r["obs"] and np.array_equal(r2["state"], st)

