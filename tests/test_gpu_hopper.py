"""Hopper kernels against the oracle: two independent formulations of the same planar 6-DoF chain
(oracle: recursive Newton-Euler on a tree table + dense LDL; kernel: absolute-angle closed forms).
Parity with libmujoco is unpinned."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

INTEGRATORS = ["euler", "semi_implicit_euler", "rk4"]


def _engine(*a, **k):
    from emei_amd.engine import Engine

    return Engine(*a, **k)


def _states(rng, n):
    """flight, standing on the foot, pressed into the floor (several contacts), joints past their limits, lying"""
    q = rng.normal(0, 0.1, (n, 6))
    q[:, 1] = 1.25 + rng.uniform(-0.08, 0.3, n)
    q[:, 3:5] = -np.abs(rng.normal(0, 0.3, (n, 2)))
    k = n // 5
    q[:k, 3:] = rng.uniform(-3.0, 0.5, (k, 3))           # limits
    q[k : 2 * k, 1] = rng.uniform(0.0, 0.3, k)           # lying / deep contact
    q[k : 2 * k, 2] = rng.uniform(-1.7, 1.7, k)
    v = rng.normal(0, 1.5, (n, 6))
    return np.concatenate([q, v], axis=1)


@pytest.mark.parametrize("integrator", INTEGRATORS)
@pytest.mark.parametrize("fr,dt", [(1, 0.002), (4, 0.002), (2, 0.01)])
@pytest.mark.parametrize("solver", ["newton", "sweep1"])
def test_onestep_vs_oracle(integrator, fr, dt, solver):
    from oracle import oracle as O

    rng = np.random.default_rng(3)
    n = 1000  # ragged last wave
    s0 = _states(rng, n)
    act = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)
    eng = _engine("HopperRunning", n, freq_rate=fr, real_time_scale=dt, integrator=integrator, solver=solver)
    eng.set_state(s0)
    obs, rew, done = eng.step(torch.as_tensor(act, device=eng.device))
    o_st, o_rew, o_term = O.hopper_step(s0, act.astype(np.float64), fr, dt, O.opts(integrator, solver=solver))
    assert rel_err(eng.get_state().cpu().numpy(), o_st, floor=1.0) <= 1e-9
    assert rel_err(obs.cpu().numpy(), o_st) <= 1e-5
    assert rel_err(rew.cpu().numpy(), o_rew) <= 1e-5
    assert not done.any() and not o_term.any()  # hopper.py:104-106: never terminal


@pytest.mark.parametrize("integrator,solver", [("rk4", "newton"), ("euler", "newton"), ("rk4", "sweep1"), ("euler", "sweep1")])
def test_capsule_pairs_vs_oracle(integrator, solver):
    """hopper.xml:5: the geoms collide with each other (torso-leg, torso-foot, thigh-foot).  Folded legs in the air and on the
    ground: the oracle sees pair rows in most of these states, several states have all three kinds of row at once, and the
    kernels' next state agrees to 1e-9."""
    from oracle import oracle as O

    rng = np.random.default_rng(21)
    n = 1536
    q = np.concatenate([rng.normal(0, 0.3, (n, 1)), rng.uniform(0.1, 2.5, (n, 1)), rng.normal(0, 1.2, (n, 1)),
                        rng.uniform(-2.9, -0.9, (n, 1)), rng.uniform(-2.9, -1.4, (n, 1)), rng.uniform(-0.9, 0.9, (n, 1))], axis=1)
    s0 = np.concatenate([q, rng.normal(0, 2.0, (n, 6))], axis=1)
    mask = O.planar_row_mask("hopper", s0)
    pairs = (mask >> 11) & 7
    assert (pairs != 0).mean() > 0.4 and all(((pairs >> b) & 1).sum() >= 5 for b in range(3)), [(pairs >> b & 1).sum() for b in range(3)]
    assert ((pairs != 0) & ((mask & 7) != 0) & (((mask >> 3) & 0xFF) != 0)).sum() >= 20  # pair + limit + floor rows together
    act = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
    eng = _engine("HopperRunning", n, freq_rate=4, real_time_scale=0.002, integrator=integrator, solver=solver)
    eng.set_state(s0)
    eng.step(torch.as_tensor(act, device=eng.device))
    o_st, _, _ = O.hopper_step(s0, act.astype(np.float64), 4, 0.002, O.opts(integrator, solver=solver))
    got = eng.get_state().cpu().numpy()
    assert rel_err(got, o_st, floor=1.0) <= 1e-9
    assert eng.solver_cap_hits() == 0


def test_f32_mode_tracks_oracle():
    from oracle import oracle as O

    rng = np.random.default_rng(4)
    n = 512
    s0 = _states(rng, n)[(np.arange(n) % 5) >= 2]  # the stiff limit / deep-contact rows amplify float32 rounding
    act = rng.uniform(-1, 1, (len(s0), 3)).astype(np.float32)
    eng = _engine("HopperRunning", len(s0), freq_rate=4, real_time_scale=0.002, integrator="rk4", precision="f32")
    eng.set_state(s0)
    obs, rew, _ = eng.step(torch.as_tensor(act, device=eng.device))
    o_st, o_rew, _ = O.hopper_step(s0, act.astype(np.float64), 4, 0.002, O.opts("rk4"))
    assert rel_err(obs.cpu().numpy(), o_st) <= 5e-3


def test_rollout_segments_vs_oracle_and_step_equivalence():
    from oracle import oracle as O

    rng = np.random.default_rng(4)
    n, T = 128, 60
    acts = rng.uniform(-1, 1, (T, n, 3)).astype(np.float32)
    kw = dict(freq_rate=4, real_time_scale=0.002, init_noise=5e-3, integrator="rk4", seed=5)
    a, b = _engine("HopperRunning", n, **kw), _engine("HopperRunning", n, **kw)
    a.reset(5)
    b.reset(5)
    s = a.get_state().cpu().numpy()
    assert np.abs(s[:, 1] - 1.25).max() < 0.03 and s[:, 1].std() == pytest.approx(5e-3, rel=0.3)  # init_qpos[rootz] = 1.25
    for e in (0, 77):
        assert rel_err(s[e], O.body_init(5, e, 0, 6, 5e-3) + np.array([0, 1.25] + [0] * 10), floor=1e-3) <= 1e-5
    dev = torch.as_tensor(acts, device=a.device)
    for t0 in range(0, T, 20):  # re-synchronise the oracle every 20 steps (80 RK4 substeps): chaotic contacts
        st = a.get_state().cpu().numpy()
        obs, rew, done = a.rollout(dev[t0 : t0 + 20].contiguous())
        for t in range(20):
            o, r, d = b.step(dev[t0 + t])
            assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t])
            st, o_rew, _ = O.hopper_step(st, acts[t0 + t].astype(np.float64), 4, 0.002, O.opts("rk4"))
            assert rel_err(obs[t].cpu().numpy(), st) <= 1e-5, (t0, t)
            assert rel_err(rew[t].cpu().numpy(), o_rew) <= 1e-4, (t0, t)
    assert torch.equal(a.get_state(), b.get_state())


def test_reference_tests_and_env_api(hopper_golden):
    """test/test_envs/test_mujoco/test_hopper.py of the reference, run against the HIP engine."""
    import emei_amd
    from emei_amd import engine as E

    env = emei_amd.HopperRunningEnv()
    assert env.is_healthy(np.ones([128, 12])).shape == (128,) and np.all(env.is_healthy(np.ones([128, 12])))  # :9-10
    assert not np.any(env.is_healthy(np.ones([128, 12]) * 101))                                            # :12-13
    reward = env.get_batch_reward(obs=np.ones([128, 12]), pre_obs=np.ones([128, 12]), action=np.ones([128, 3]))
    assert reward.shape == (128, 1) and np.allclose(reward, 1.0 - 3e-3)                                    # :15-20
    assert env.get_batch_terminal(obs=np.ones([128, 12])).shape == (128, 1)                                # :22-25
    np.random.seed(0)
    obs, info = env.reset()
    assert obs.shape == (12,) and abs(obs[1] - 1.25) < 0.05                                                # :30-31
    obs, reward, terminal, truncated, info = env.step(env.action_space.sample())
    assert obs.shape == (12,) and not terminal and np.isfinite(reward)                                     # :33-35
    for prm in (0.01, {0: (0.1, 0.1)}):                                                                    # :38-53
        env = emei_amd.HopperRunningEnv(obs_noise_params=prm)
        obs, info = env.reset()
        obs, reward, terminal, truncated, info = env.step(env.action_space.sample())
        assert obs.shape == (12,) and np.isfinite(obs).all()
    g = hopper_golden
    o, po, ac = g["hopper_obs"], g["hopper_pre_obs"], g["hopper_action"]
    r = env.get_batch_reward(o, po, ac)  # float64 rows, every golden row
    assert rel_err(r[:, 0], g["hopper_reward_B1"]) <= 1e-5
    assert np.array_equal(env.get_batch_terminal(o), g["hopper_terminal"]) and not g["hopper_terminal"].any()
    t = E.batch_terminal("HopperRunning", torch.as_tensor(np.nan_to_num(o), dtype=torch.float32, device="cuda")).cpu().numpy()
    assert not t.any()
    # the whole-batch control cost hopper.py:98 executes for B > 1 (np.sum without an axis): opt-in
    assert rel_err(env.get_batch_reward(o, po, ac, reference_batch_semantics=True), g["hopper_reward_batchquirk"]) <= 1e-5


def test_auto_reset_truncation_and_dataset():
    """TimeLimit 1000 of register_env.py:87-91 realised on the device; auto-reset re-draws the init state."""
    import emei_amd
    from emei_amd import datasets

    env = emei_amd.make("HopperRunning-v0", num_envs=256, max_episode_steps=5, auto_reset=True)
    env.reset(seed=3, options={"device_rng": True})
    acts = (torch.rand((12, 256, 3), device="cuda") * 2 - 1).float()
    obs, rew, term, trunc = env.rollout(acts)
    assert not term.any() and trunc[4].all() and trunc[9].all() and trunc.sum() == 2 * 256
    st = env.engine.get_state().cpu().numpy()
    assert np.isfinite(st).all() and np.abs(st[:, 1] - 1.25).max() < 0.1  # 2 steps after the second reset
    d, info = datasets.collect(emei_amd.make("HopperRunning-v0", num_envs=64, max_episode_steps=5, auto_reset=True), 12, seed=1)
    assert d["observations"].shape == (12 * 64, 12) and d["actions"].shape == (12 * 64, 3)
    assert int(d["timeouts"].sum()) == 2 * 64 and torch.equal(d["dones"], d["timeouts"]) and info["total_episode_num"] == 128
    # observations after a reset are the device init obs: z back near 1.25
    o = d["observations"].reshape(64, 12, 12)
    assert float((o[:, 5, 1] - 1.25).abs().max()) < 0.03 and float((o[:, 10, 1] - 1.25).abs().max()) < 0.03


def test_constructor_parameters_on_the_device(hopper_golden, mujoco_golden):
    """Non-default reward / health parameters (hopper.py:25-30, half_cheetah.py:23-24) travel as emei_config.env_params:
    stateless functions vs the golden vectors, fused step vs the oracle, and a Hopper that really terminates."""
    import emei_amd
    from oracle import oracle as O

    g = hopper_golden
    o, po, ac = g["hopper_obs"], g["hopper_pre_obs"], g["hopper_action"]
    env = emei_amd.HopperRunningEnv(forward_reward_weight=2.0, ctrl_cost_weight=5e-3, healthy_reward=0.5, terminate_when_unhealthy=False,
                                    healthy_state_range=(-50.0, 60.0), healthy_z_range=(0.8, 1.5), num_envs=512, auto_reset=True)
    assert np.array_equal(env.get_batch_terminal(o), g["hopper_custom_terminal"])  # every row: float64 z against the float64 thresholds
    assert rel_err(env.get_batch_reward(o, po, ac)[:, 0], g["hopper_custom_reward_B1"]) <= 1e-5
    prm = dict(zip(O.ENV_PARAM_ORDER, g["hopper_custom_params"]))
    rng = np.random.default_rng(6)
    q = rng.normal(0, 0.1, (512, 6))
    q[:, 1] = rng.uniform(0.5, 1.7, 512)  # around both z thresholds
    s0 = np.concatenate([q, rng.normal(0, 1, (512, 6))], axis=1)
    act = rng.uniform(-1, 1, (512, 3)).astype(np.float32)
    env.engine.set_state(s0)
    obs, rew, done = env.engine.step(torch.as_tensor(act, device="cuda"), auto_reset=False)
    o_st, o_rew, o_term = O.hopper_step(s0, act.astype(np.float64), 4, 0.002, O.opts("rk4"), prm)
    near = (np.abs(o_st[:, 1] - 0.8) < 1e-9) | (np.abs(o_st[:, 1] - 1.5) < 1e-9)
    assert rel_err(obs.cpu().numpy(), o_st) <= 1e-5 and rel_err(rew.cpu().numpy()[~near], o_rew[~near]) <= 2e-5
    assert np.array_equal((done.cpu().numpy() & 1).astype(bool)[~near], o_term[~near]) and o_term.any() and not o_term.all()
    # episodes now end: a rollout with auto-reset counts terminal steps and keeps z inside the range right after resets
    env.reset(seed=1, options={"device_rng": True})
    acts = (torch.rand((60, 512, 3), device="cuda") * 2 - 1).float()
    _, _, term, trunc = env.rollout(acts)
    assert bool(term.any()) and not bool(trunc.any())
    # cheetah weights
    m = mujoco_golden
    ch = emei_amd.HalfCheetahRunningEnv(forward_reward_weight=2.5, ctrl_cost_weight=0.03)
    r = ch.get_batch_reward(m["cheetah_obs"], m["cheetah_pre_obs"], m["cheetah_action"])
    assert rel_err(r[:, 0], m["cheetah_reward_B1_w2p5_c0p03"]) <= 1e-5
    with pytest.raises(NotImplementedError):  # envs without such parameters refuse them
        from emei_amd.engine import Engine

        Engine("CartPoleSwingUp", 64, env_params={"forward_reward_weight": 2.0})
