"""ctypes binding of libemei_hip.so (C ABI: include/emei_hip.h).

The HIP library is the product path: there is no CPU or PyTorch fallback.  If the shared object is
missing or does not export the ABI, importing this module's `lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EMEI_HIP_LIB lets a developer A/B two builds of the same ABI inside one process launch; the default
# is the in-tree library next to this file.
LIB_PATH = os.environ.get("EMEI_HIP_LIB") or os.path.join(_HERE, "libemei_hip.so")

# enum emei_env_id
ENV_IDS = {
    "CartPoleSwingUp": 0,
    "CartPoleBalancing": 1,
    "ReboundInvertedPendulumBalancing": 2,
    "BoundaryInvertedPendulumBalancing": 3,
    "ReboundInvertedPendulumSwingUp": 4,
    "BoundaryInvertedPendulumSwingUp": 5,
    "HalfCheetahRunning": 6,
    "ReboundInvertedDoublePendulumBalancing": 7,
    "BoundaryInvertedDoublePendulumBalancing": 8,
    "ReboundInvertedDoublePendulumSwingUp": 9,
    "BoundaryInvertedDoublePendulumSwingUp": 10,
    "HopperRunning": 11,
}
PRECISION_REF, PRECISION_F32 = 0, 1
ACT_U8, ACT_I32, ACT_I64, ACT_F32 = 0, 1, 2, 3
FLAG_AUTO_RESET = 1
DONE_TERMINAL, DONE_TRUNCATED = 1, 2
OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4
ABI_VERSION = 7
IO_F32, IO_F64 = 0, 1  # enum emei_io_dtype
REWARD_BATCH_CTRL_COST = 1  # flag of emei_reward_io (half_cheetah.py:61 / hopper.py:98: np.sum over the whole batch)
# enum emei_kernel_id (emei_last_rollout_kernel)
KERNEL_NAMES = {0: "none", 1: "pend_rollout_staged_kernel<freq1>", 2: "pend_rollout_staged_kernel", 3: "pend_rollout_kernel<full>",
                4: "pend_rollout_kernel", 5: "body_rollout_kernel", 6: "body_rollout_kernel<rk4>",
                7: "body_rollout_kernel (chunked work items)", 8: "body_rollout_kernel<rk4> (chunked work items)",
                9: "pend_rollout_staged_peers_kernel<freq1>", 10: "pend_rollout_staged_peers_kernel"}
KERNEL_PEND_STAGED_FREQ1, KERNEL_PEND_STAGED, KERNEL_PEND_GENERIC_FULL, KERNEL_PEND_GENERIC, KERNEL_BODY, KERNEL_BODY_RK4 = 1, 2, 3, 4, 5, 6
KERNEL_BODY_CHUNKED, KERNEL_BODY_RK4_CHUNKED = 7, 8
KERNEL_PEND_STAGED_PEERS_FREQ1, KERNEL_PEND_STAGED_PEERS = 9, 10
MAX_OBS_PEERS = 8  # EMEI_MAX_OBS_PEERS
# enum emei_ode_method: `method` of ODE_approximation (base_control.py:133-173), classic control only
ODE_METHODS = {"euler": 0, "rk4": 1}
NEXT_OBS_ODE_RK4 = 0x100  # flag on emei_next_obs_io's `integrator`
# enum emei_integrator (mujoco_env.py:70-79) / enum emei_noise_layout
INTEGRATORS = {"euler": 0, "semi_implicit_euler": 1, "rk4": 2}
NOISE_IID, NOISE_SHARED = 0, 1
# enum emei_solver
SOLVERS = {"newton": 0, "sweep1": 1}
CONFIG_SIZE_V1 = 64
MAX_STATE_DIM = 32
MAX_ENV_PARAMS = 8
# enum emei_env_param
ENV_PARAMS = {"forward_reward_weight": 0, "ctrl_cost_weight": 1, "healthy_reward": 2, "terminate_when_unhealthy": 3,
              "healthy_state_lo": 4, "healthy_state_hi": 5, "healthy_z_lo": 6, "healthy_z_hi": 7}


class EmeiConfig(C.Structure):
    """struct emei_config (include/emei_hip.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("env_id", C.c_int32),
        ("n_envs", C.c_int64),
        ("freq_rate", C.c_int32),
        ("precision", C.c_int32),
        ("real_time_scale", C.c_double),
        ("max_episode_steps", C.c_int32),
        ("device", C.c_int32),
        ("seed", C.c_uint64),
        ("env_index_offset", C.c_uint64),
        ("init_noise", C.c_double),
        # ABI version 2
        ("integrator", C.c_int32),
        ("noise_layout", C.c_int32),
        ("init_sigma", C.c_float * MAX_STATE_DIM),
        ("obs_sigma", C.c_float * MAX_STATE_DIM),
        # struct_size 400: constructor parameters of the reward / terminal functions
        ("env_param_mask", C.c_uint32),
        ("solver", C.c_uint32),
        ("env_params", C.c_double * MAX_ENV_PARAMS),
        # struct_size 408 (ABI 6)
        ("ode_method", C.c_int32),
        ("rollout_chunk_steps", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/emei_hip.h declares
_vp, _i32, _i64, _u32, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
SYMBOLS = {
    "emei_create": (C.c_int, [C.POINTER(EmeiConfig), C.POINTER(_vp)]),
    "emei_destroy": (C.c_int, [_vp]),
    "emei_last_error": (C.c_char_p, []),
    "emei_abi_version": (C.c_int, []),
    "emei_model_constants": (C.c_int, [C.c_int, _vp, C.c_int]),
    "emei_model_invweights": (C.c_int, [C.c_int, _vp, C.c_int]),
    "emei_env_dims": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "emei_reset": (C.c_int, [_vp, _u64, _vp]),
    "emei_set_seed": (C.c_int, [_vp, _u64]),
    "emei_get_solver_cap_hits": (C.c_int, [_vp, _vp, _vp]),
    "emei_get_rollout_faults": (C.c_int, [_vp, _vp, _vp]),
    "emei_last_rollout_kernel": (C.c_int, [_vp]),
    "emei_set_obs_peers": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), _i64, _i64, _i32]),
    "emei_peer_buffer_create": (C.c_int, [C.c_int, _u64, C.POINTER(_vp), _vp]),
    "emei_peer_buffer_open": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "emei_peer_buffer_close": (C.c_int, [C.c_int, _vp]),
    "emei_peer_buffer_destroy": (C.c_int, [C.c_int, _vp]),
    "emei_set_state": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "emei_get_state": (C.c_int, [_vp, _vp, _vp]),
    "emei_get_obs": (C.c_int, [_vp, _vp, _vp]),
    "emei_freeze": (C.c_int, [_vp, _vp]),
    "emei_unfreeze": (C.c_int, [_vp, _vp]),
    "emei_step": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _u32, _vp]),
    "emei_step_host": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _u32, _vp]),
    "emei_rollout": (C.c_int, [_vp, _i32, _vp, C.c_int, _vp, _vp, _vp, _u32, _vp]),
    "emei_compact_done": (C.c_int, [_vp, _vp, _vp, _vp]),
    "emei_get_counters": (C.c_int, [_vp, _vp, _vp, _vp]),
    "emei_episode_init_obs": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "emei_reward": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp, _dbl, _i32, _vp, _vp]),
    "emei_terminal": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp]),
    "emei_reward_ex": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp, _dbl, _i32, _u32, _vp, _vp, _vp]),
    "emei_terminal_ex": (C.c_int, [C.c_int, _i64, _vp, _u32, _vp, _vp, _vp]),
    "emei_next_obs": (C.c_int, [C.c_int, _i64, _vp, _vp, C.c_int, _dbl, _i32, _i32, _vp, _vp]),
    "emei_next_obs_ex": (C.c_int, [C.c_int, _i64, _vp, _vp, C.c_int, _dbl, _i32, _i32, _i32, _vp, _vp]),
    "emei_reward_io": (C.c_int, [C.c_int, _i64, C.c_int, _vp, _vp, _vp, _dbl, _i32, _u32, _vp, _u32, _vp, _vp]),
    "emei_terminal_io": (C.c_int, [C.c_int, _i64, C.c_int, _vp, _u32, _vp, _vp, _vp]),
    "emei_next_obs_io": (C.c_int, [C.c_int, _i64, C.c_int, _vp, _vp, C.c_int, _dbl, _i32, _i32, _i32, _vp, _vp]),
}

_lib = None


class EmeiHipError(RuntimeError):
    pass


def lib():
    """Load libemei_hip.so; fail loudly (no fallback) when it is absent or stale."""
    global _lib
    if _lib is None:
        # torch must be imported first: its bundled HIP runtime and /opt/rocm's share one SONAME, and
        # whichever loads first serves both; loading ours before torch leaves torch without a device.
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise EmeiHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C emei_amd/csrc`). emei_amd has no CPU fallback."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the ABI symbol is not exported
            fn.restype, fn.argtypes = res, args
        if l.emei_abi_version() != ABI_VERSION:
            raise EmeiHipError(f"libemei_hip.so ABI {l.emei_abi_version()} != binding {ABI_VERSION}")
        _lib = l
    return _lib


def pack_env_params(params):
    """{name: value} (names of ENV_PARAMS) -> (mask, c_double[MAX_ENV_PARAMS]); None / {} -> the defaults."""
    arr = (C.c_double * MAX_ENV_PARAMS)()
    mask = 0
    for k, v in (params or {}).items():
        idx = ENV_PARAMS[k]
        arr[idx] = float(v)
        mask |= 1 << idx
    return mask, arr


def check(rc):
    """Map ABI status codes onto the exceptions the reference raises at the same places."""
    if rc == OK:
        return
    msg = lib().emei_last_error().decode()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_STATE:
        raise AssertionError(msg)  # `assert self.state is not None`, base_control.py:67
    raise EmeiHipError(msg)
