"""Host-side mirror of the reference's API surface: everything here runs without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import emei_amd
from emei_amd import _lib, spaces
from emei_amd.core import EmeiEnv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_env_params_name(mujoco_golden):
    """test/test_core.py:9-14 of the reference."""
    env = EmeiEnv(env_params={"a": 3, "b": 5, "d": 0.33, "c": "c"})
    assert env.env_params_name == "a=3&b=5&c=c&d=0.33"
    env = EmeiEnv(env_params=dict(freq_rate=1, real_time_scale=0.02))
    assert env.env_params_name == "freq_rate=1&real_time_scale=0.02"
    names = [str(x) for x in mujoco_golden["env_params_names"]]
    assert names[0] == "a=3&b=5&c=c&d=0.33"
    assert emei_amd.HalfCheetahRunningEnv().env_params_name == names[2]
    assert emei_amd.CartPoleSwingUpEnv().env_name == "CartPoleSwingUp"


def test_freeze_flags():
    """test/test_core.py:17-23 of the reference."""
    env = EmeiEnv(env_params=dict(freq_rate=1, time_step=0.02))
    env.freeze()
    assert env.frozen
    with pytest.raises(AssertionError):
        env.freeze()
    env.unfreeze()
    assert not env.frozen
    with pytest.raises(AssertionError):
        env.unfreeze()


def test_abstract_cartpole_reset_raises():
    """test_cartpole.py:4-11 of the reference."""
    with pytest.raises(NotImplementedError):
        emei_amd.BaseCartPoleEnv().reset()


def test_transition_graph_closure(mujoco_golden):
    env = emei_amd.BoundaryInvertedPendulumSwingUpEnv()
    for rt in (1, 2, 3, 5):
        assert np.array_equal(env.get_transition_graph(rt), mujoco_golden[f"ip_graph_repeat{rt}"])
    with pytest.raises(AttributeError):  # CartPole defines no graph: None.copy() (core.py:143)
        emei_amd.CartPoleSwingUpEnv().get_transition_graph()


@pytest.mark.parametrize("name,cls", [("swingup", emei_amd.CartPoleSwingUpEnv), ("balancing", emei_amd.CartPoleBalancingEnv)])
def test_host_init_states_match_reference(cartpole_golden, name, cls):
    g = cartpole_golden
    for seed in range(16):
        env = cls()
        env._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        assert np.array_equal(env.get_batch_init_state(1)[0], g[f"reset_{name}_seeds0_15"][seed])
    # the fixture was drawn after reset(seed=7), which itself consumes one row (base_control.py:44-46)
    env._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(7)))
    env.get_batch_init_state(1)
    assert np.array_equal(env.get_batch_init_obs(8), g[f"batchinitobs_{name}_seed7_B8"])


def test_mujoco_init_noise_quirk(mujoco_golden):
    """mujoco_env.py:243-244: for B = 1 one draw is added to every qpos entry, one to every qvel entry."""
    g = mujoco_golden
    for seed in range(8):
        np.random.seed(seed)
        pos, vel = emei_amd.BoundaryInvertedPendulumBalancingEnv().get_batch_init_state(1)
        assert np.array_equal(pos[0], g["noise_ip_pos_seeds0_7"][seed]) and np.array_equal(vel[0], g["noise_ip_vel_seeds0_7"][seed])
        np.random.seed(seed)
        pos, vel = emei_amd.HalfCheetahRunningEnv().get_batch_init_state(1)
        assert np.array_equal(pos[0], g["noise_cheetah_pos_seeds0_7"][seed]) and np.array_equal(vel[0], g["noise_cheetah_vel_seeds0_7"][seed])
    pos, vel = emei_amd.BoundaryInvertedPendulumBalancingEnv().get_batch_init_state(64)
    assert pos.shape == (64, 2) and vel.shape == (64, 2) and np.unique(pos).size == 128  # i.i.d. per coordinate


def test_spaces_and_registry():
    d = spaces.Discrete(2)
    assert d.contains(0) and d.contains(np.asarray(1)) and not d.contains(2) and not d.contains(0.5) and not d.contains(np.asarray([1]))
    b = spaces.Box(-3.0, 3.0, shape=(1,), dtype=np.float32)
    assert b.contains(np.array([2.5], np.float32)) and not b.contains(np.array([3.5], np.float32)) and not b.contains(np.zeros(2))
    assert emei_amd.spec("CartPoleSwingUp-v0")["max_episode_steps"] == 1000  # register_env.py:19-23
    assert emei_amd.spec("CartPoleBalancing-v0")["max_episode_steps"] == 500  # register_env.py:14-18
    env = emei_amd.make("BoundaryInvertedPendulumBalancing-v0", freq_rate=4)
    assert env.max_episode_steps == 1000 and env.freq_rate == 4 and env.observation_space.shape == (4,)
    assert emei_amd.make("HalfCheetahRunning-v0").dt == pytest.approx(0.008)
    assert emei_amd.spec("HopperRunning-v0")["max_episode_steps"] == 1000  # register_env.py:87-91
    hop = emei_amd.make("HopperRunning-v0")
    assert hop.integrator == "rk4" and hop.dt == pytest.approx(0.008) and hop.action_space.shape == (3,)  # hopper.py:20-22
    with pytest.raises(KeyError):
        emei_amd.spec("Walker2dRunning-v0")  # registered by the reference, class never written (register_env.py:92-96)
    with pytest.raises(NotImplementedError):  # mujoco_env.py:78-79
        emei_amd.BoundaryInvertedPendulumBalancingEnv(integrator="verlet")
    assert emei_amd.make("CartPoleSwingUp-v0", render_mode=None).render_mode is None  # base_control.py:15
    with pytest.raises(NotImplementedError):
        emei_amd.make("HopperRunning-v0", render_mode="human")
    for integ in ("euler", "semi_implicit_euler", "rk4"):
        e = emei_amd.BoundaryInvertedPendulumBalancingEnv(integrator=integ, obs_noise_params=(1e-3, 2e-3))
        assert e.env_params_name == f"freq_rate=1&integrator={integ}&real_time_scale=0.02"  # mujoco_env.py:51
    emei_amd.HopperRunningEnv(obs_noise_params={0: (0.1, 0.1)})  # the dict form of test_hopper.py:47


def test_abi_library_exports_every_declared_symbol():
    """include/emei_hip.h is the contract: every EMEI_API function must be exported by the .so
    (no compute call is made here)."""
    hdr = open(os.path.join(ROOT, "include", "emei_hip.h")).read()
    declared = set(re.findall(r"EMEI_API\s+[\w\s\*]+?\b(emei_\w+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.emei_abi_version() == _lib.ABI_VERSION
    od, ad, sd = C.c_int(), C.c_int(), C.c_int()
    assert lib.emei_env_dims(_lib.ENV_IDS["HalfCheetahRunning"], C.byref(od), C.byref(ad), C.byref(sd)) == 0
    assert (od.value, ad.value, sd.value) == (18, 6, 18)
    assert lib.emei_env_dims(99, None, None, None) == _lib.ERR_INVALID
    # the struct layout of the binding is checked by the library itself
    cfg = _lib.EmeiConfig(C.sizeof(_lib.EmeiConfig) - 4, 0, 16, 1, 0, 0.02, 0, 0, 0, 0, 0.0)  # (- 8 is the ABI-5 size: accepted)
    h = C.c_void_p()
    assert lib.emei_create(C.byref(cfg), C.byref(h)) == _lib.ERR_INVALID
    assert b"emei_config size" in lib.emei_last_error()
    with pytest.raises(ValueError):
        _lib.check(_lib.ERR_INVALID)
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("emei_")} == declared  # nothing undeclared leaks out either


def test_create_rejects_bad_configs_before_touching_a_device():
    """Argument validation of emei_create happens on the host, ahead of any HIP call: sizes the kernels' 32-bit env index
    cannot address, an unknown solver / integrator / precision, negative sigmas.  (No GPU needed: nothing is launched.)"""
    lib = _lib.lib()

    def create(**kw):
        cfg = _lib.EmeiConfig()
        cfg.struct_size = C.sizeof(_lib.EmeiConfig)
        cfg.env_id, cfg.n_envs, cfg.freq_rate, cfg.real_time_scale = _lib.ENV_IDS["HalfCheetahRunning"], 64, 1, 0.02
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.emei_create(C.byref(cfg), C.byref(h))
        assert not h.value
        return rc, lib.emei_last_error().decode()

    for bad in (0, -5, 1 << 31, 1 << 40):
        rc, msg = create(n_envs=bad)
        assert rc == _lib.ERR_INVALID and "n_envs" in msg, (bad, msg)
    assert create(solver=2) == (_lib.ERR_INVALID, "emei_create: solver=2")
    assert create(integrator=3)[0] == _lib.ERR_UNSUPPORTED  # mujoco_env.py:78-79 raises NotImplementedError
    assert create(precision=7)[0] == _lib.ERR_INVALID
    assert create(freq_rate=0)[0] == _lib.ERR_INVALID
    assert create(real_time_scale=0.0)[0] == _lib.ERR_INVALID
    assert create(real_time_scale=float("nan"))[0] == _lib.ERR_INVALID
    assert create(noise_layout=2)[0] == _lib.ERR_INVALID
    assert create(ode_method=2)[0] == _lib.ERR_UNSUPPORTED  # base_control.py:171-172 raises NotImplementedError
    rc, msg = create(ode_method=_lib.ODE_METHODS["rk4"])  # ODE_approximation is classic control's; the bodies switch on `integrator`
    assert rc == _lib.ERR_INVALID and "ode_method" in msg
    assert create(rollout_chunk_steps=-2)[0] == _lib.ERR_INVALID
    assert create(env_id=99)[0] == _lib.ERR_INVALID
    assert create(env_param_mask=1 << 8)[0] == _lib.ERR_INVALID
    assert create(env_param_mask=1 << 2)[0] == _lib.ERR_INVALID  # HalfCheetah has only the two reward weights
    assert create(env_id=_lib.ENV_IDS["CartPoleSwingUp"], env_param_mask=1)[0] == _lib.ERR_UNSUPPORTED
    sig = (C.c_float * 32)(*([0.0] * 32))
    sig[3] = -1e-3
    assert create(init_sigma=sig)[0] == _lib.ERR_INVALID


def test_no_product_import_of_the_oracle():
    """The product package must never reach into oracle/ (it is test infrastructure); neither do the developer tools —
    scripts that check against the oracle live under tests/host/."""
    for top in ("emei_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in txt and "from oracle" not in txt and "libemei_oracle" not in txt, (top, f)


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EmeiHipError):
        emei_amd.CartPoleSwingUpEnv().reset(seed=0)


def test_offline_dataset_lookup_is_local_only(tmp_path, monkeypatch):
    """core.py:82-128: datasets live under <root>/<env_name>-v0/<env_params_name>/<name>.h5 (the path the reference's URL scheme
    resolves to, in the reference's HDF5 container) or, rounds 1-3, <root>/<env_name>/<params>/<name>.npz; never downloaded."""
    from emei_amd import core, datasets, h5io

    monkeypatch.setattr(core, "DATASET_PATH", tmp_path)
    env = emei_amd.CartPoleSwingUpEnv(freq_rate=2)
    params = "freq_rate=2&integrator=euler&real_time_scale=0.02"
    assert env.dataset_dir == tmp_path / "CartPoleSwingUp" / params
    assert env.reference_dataset_dir == tmp_path / "CartPoleSwingUp-v0" / params
    assert env.dataset_names == []
    with pytest.raises(AssertionError):  # `assert dataset_name in self._offline_dataset_urls`, core.py:110
        env.get_dataset("uniform")
    n = 32
    data = {"observations": np.zeros((n, 4), np.float32), "next_observations": np.ones((n, 4), np.float32),
            "actions": np.zeros((n, 1), np.float32), "rewards": np.zeros(n, np.float32), "dones": np.zeros(n, np.float32),
            "timeouts": np.zeros(n, np.float32)}
    path = datasets.save_for_env(env, data, "uniform", rollout_info={"total_sample_num": n})
    assert path == env.reference_dataset_dir / "uniform.h5" and path.exists() and env.dataset_names == ["uniform"]
    assert path.read_bytes()[:8] == h5io.SIGNATURE
    got = env.get_dataset("uniform")
    assert set(datasets.DATASET_KEYS) == set(got) and np.array_equal(got["next_observations"], data["next_observations"])
    assert set(env.load_h5_data(path)) == set(datasets.DATASET_KEYS)  # core.py:61-81
    old = datasets.save_for_env(env, data, "legacy", fmt="npz")
    assert old == env.dataset_dir / "legacy.npz" and env.dataset_names == ["legacy", "uniform"]
    assert np.array_equal(env.get_dataset("legacy")["observations"], data["observations"])
    bad = dict(data)
    del bad["timeouts"]
    for fmt in ("h5", "npz"):
        datasets.save_for_env(env, bad, "broken", fmt=fmt)
        with pytest.raises(AssertionError, match="Dataset is missing key timeouts"):  # core.py:118-126
            env.get_dataset("broken")
        (env.reference_dataset_dir / "broken.h5").unlink(missing_ok=True)


def test_mujoco_backed_envs_warn_that_parity_is_unpinned():
    """the MuJoCo-backed env classes present themselves as drop-in equivalents, so constructing one says — once
    per class — that its next-state values are not pinned against libmujoco."""
    import warnings

    import emei_amd
    from emei_amd.envs.base import MujocoHipEnv, ParityUnpinnedWarning

    MujocoHipEnv._warned.discard("HopperRunningEnv")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        emei_amd.HopperRunningEnv()
        emei_amd.HopperRunningEnv()
        emei_amd.CartPoleSwingUpEnv()  # first-party dynamics, pinned: no warning
    got = [x for x in w if issubclass(x.category, ParityUnpinnedWarning)]
    assert len(got) == 1 and "libmujoco" in str(got[0].message)


def test_free_joint_branch_of_the_euler_position_rule_vs_golden():
    """mujoco_env.py:176-184 (VERDICT r03 missing #5): no env on the path has a free joint, so this is host code only — the
    reference's composition of SciPy rotations (scalar-first qpos read as scalar-last, degrees + radians, zyx out / xyz in),
    restated in NumPy, against vectors made by the reference itself (oracle/gen_golden.py:gen_freejoint)."""
    import os

    from emei_amd.envs.base import check_noise_joints, euler_position_rule

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "freejoint_golden.npz"))
    for nm, jt in (("free", [0]), ("free_hinge2", [0, 3, 3]), ("slide_free_hinge", [2, 0, 3])):
        for dt in (0.02, 0.002):
            qp, qv, want = g[f"fj_{nm}_dt{dt}_qpos"], g[f"fj_{nm}_dt{dt}_qvel"], g[f"fj_{nm}_dt{dt}_newpos"]
            got = np.stack([euler_position_rule(qp[i], qv[i], jt, dt) for i in range(len(qp))])
            assert np.abs(got - want).max() <= 1e-13, (nm, dt)
            assert (np.abs(np.linalg.norm(got[:, jt.index(0) + 3: jt.index(0) + 7], axis=1) - 1) < 1e-14).all()  # unit quaternions out
    assert str(g["fj_ball_raises"]) == str(g["fj_hinge_ball_raises"]) == "NotImplementedError"
    with pytest.raises(NotImplementedError):
        euler_position_rule(np.zeros(5), np.zeros(4), [3, 1], 0.02)
    assert str(g["fj_noise_B1_raises"]) == str(g["fj_noise_B4_raises"]) == "ValueError"
    with pytest.raises(ValueError):
        check_noise_joints([0, 3])
    # the envs of this package: 1-dof joints, q += dt v (the rule the kernels apply per substep)
    env = emei_amd.HopperRunningEnv()
    q, v = np.arange(6.0), np.ones(6)
    assert np.array_equal(env.get_euler_pos(q, v), q + env.real_time_scale * v)
