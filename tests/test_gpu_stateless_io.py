"""The *_io stateless entry points (emei_reward_io / emei_terminal_io / emei_next_obs_io) beyond the golden rows: float64
get_batch_next_obs of every env family against the oracle's float64 step (1e-9: the float64 observation enters the float64
state unrounded), ragged sizes, B = 1, and the argument errors of the C ABI."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(x, dtype=torch.float64):
    return torch.as_tensor(x, dtype=dtype, device="cuda")


@pytest.mark.parametrize("n", [1, 1000])
def test_next_obs_float64_rows_vs_oracle(n):
    from emei_amd import engine as E
    from oracle import oracle as O

    rng = np.random.default_rng(12)
    # CartPole (discrete actions): float64 in, float64 out, state to 1e-9
    s = rng.uniform(-0.5, 0.5, (n, 4))
    a = rng.integers(2, size=n)
    nxt = E.batch_next_obs("CartPoleBalancing", _dev(s), torch.as_tensor(a, device="cuda"), 0.02, 2, "ref")
    assert nxt.dtype == torch.float64 and rel_err(nxt.cpu().numpy(), O.cartpole_step("balancing", s, a, 2, 0.02)[0], floor=1.0) <= 1e-8  # one float32 ulp of a derivative x dt
    # InvertedPendulum through the staged family's stateless kernel (euler) and through the Body kernel (rk4)
    s = np.column_stack([rng.uniform(-2.2, 2.2, n), rng.uniform(-3, 3, n), rng.normal(0, 2, (n, 2))])
    a = rng.uniform(-3.5, 3.5, n).astype(np.float32)
    for integ in ("euler", "rk4"):
        nxt = E.batch_next_obs("BoundaryInvertedPendulumSwingUp", _dev(s), _dev(a, torch.float32), 0.02, 4, "ref", integ)
        want = O.ip_step("boundary_swingup", s, a.astype(np.float64), 4, 0.02, O.opts(integ))[1]
        d = np.angle(np.exp(1j * (nxt.cpu().numpy()[:, 1] - want[:, 1])))  # theta wraps at +-pi
        got = nxt.cpu().numpy().copy()
        got[:, 1] = want[:, 1] + d
        assert rel_err(got, want, floor=1.0) <= 1e-9, integ
    # HalfCheetah / Hopper: contact states included
    q = rng.normal(0, 0.2, (n, 9))
    q[:, 1] = rng.uniform(-0.4, 0.2, n)
    s = np.concatenate([q, rng.normal(0, 1.5, (n, 9))], axis=1)
    a = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
    nxt = E.batch_next_obs("HalfCheetahRunning", _dev(s), _dev(a, torch.float32), 0.002, 4, "ref", "euler")
    assert rel_err(nxt.cpu().numpy(), O.cheetah_step(s, a.astype(np.float64), 4, 0.002)[0], floor=1.0) <= 1e-9
    q = rng.normal(0, 0.1, (n, 6))
    q[:, 1] = 1.25 + rng.uniform(-0.3, 0.1, n)
    s = np.concatenate([q, rng.normal(0, 1.5, (n, 6))], axis=1)
    a = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    nxt = E.batch_next_obs("HopperRunning", _dev(s), _dev(a, torch.float32), 0.002, 4, "ref", "rk4")
    assert rel_err(nxt.cpu().numpy(), O.hopper_step(s, a.astype(np.float64), 4, 0.002, O.opts("rk4"))[0], floor=1.0) <= 1e-9
    # the float32 rows of the same call round the same state: within float32 of the float64 rows
    n32 = E.batch_next_obs("HopperRunning", _dev(s, torch.float32), _dev(a, torch.float32), 0.002, 4, "ref", "rk4")
    assert n32.dtype == torch.float32 and rel_err(n32.cpu().numpy(), nxt.cpu().numpy(), floor=1.0) <= 2e-5
    with pytest.raises(NotImplementedError):  # the double pendulum's observation does not determine its state (core.py:190-193)
        E.batch_next_obs("BoundaryInvertedDoublePendulumSwingUp", _dev(np.zeros((n, 6))), _dev(np.zeros(n), torch.float32))


def test_batch_control_cost_flag_and_b1(mujoco_golden):
    from emei_amd import engine as E

    g = mujoco_golden
    o, po, ac = (_dev(np.nan_to_num(g[k])) for k in ("cheetah_obs", "cheetah_pre_obs", "cheetah_action"))
    per_env = E.batch_reward("HalfCheetahRunning", o, po, ac, 0.002, 4)
    whole = E.batch_reward("HalfCheetahRunning", o, po, ac, 0.002, 4, batch_ctrl_cost=True)
    cost = (ac**2).sum(dim=1)
    assert torch.allclose(whole, per_env + 0.1 * (cost - cost.sum()), rtol=0, atol=1e-12)
    # B = 1: the whole batch IS the row
    assert torch.equal(E.batch_reward("HalfCheetahRunning", o[:1], po[:1], ac[:1], 0.002, 4, batch_ctrl_cost=True), per_env[:1])
    # float32 rows take the same path
    w32 = E.batch_reward("HalfCheetahRunning", o.float(), po.float(), ac.float(), 0.002, 4, batch_ctrl_cost=True)
    assert w32.dtype == torch.float32 and rel_err(w32.cpu().numpy(), whole.cpu().numpy(), floor=1.0) <= 1e-5
    with pytest.raises(NotImplementedError):  # an env without a control-cost term
        E.batch_reward("CartPoleSwingUp", _dev(np.zeros((4, 4))), batch_ctrl_cost=True)


def test_c_abi_argument_errors():
    from emei_amd import _lib as L

    lib = L.lib()
    obs = torch.zeros((8, 4), dtype=torch.float64, device="cuda")
    out = torch.zeros(8, dtype=torch.float64, device="cuda")
    done = torch.zeros(8, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    z = C.c_void_p(0)
    assert lib.emei_reward_io(0, 8, 7, p(obs), z, z, 0.02, 1, 0, z, 0, p(out), z) == L.ERR_INVALID            # io_dtype
    assert lib.emei_reward_io(0, 8, L.IO_F64, p(obs), z, z, 0.02, 1, 0, z, 2, p(out), z) == L.ERR_INVALID     # unknown flag
    assert lib.emei_reward_io(0, 0, L.IO_F64, p(obs), z, z, 0.02, 1, 0, z, 0, p(out), z) == L.ERR_INVALID     # n
    assert lib.emei_reward_io(6, 8, L.IO_F64, p(obs), z, z, 0.002, 4, 0, z, 0, p(out), z) == L.ERR_INVALID    # cheetah needs pre_obs, action
    assert lib.emei_terminal_io(0, 8, L.IO_F64, C.c_void_p(obs.data_ptr() + 8), 0, z, p(done), z) == L.ERR_INVALID  # misaligned rows
    assert lib.emei_terminal_io(99, 8, L.IO_F64, p(obs), 0, z, p(done), z) == L.ERR_INVALID
    assert lib.emei_next_obs_io(0, 8, L.IO_F64, p(obs), p(done), L.ACT_U8, 0.02, 1, 5, 0, p(obs), z) == L.ERR_INVALID  # precision
    assert lib.emei_next_obs_io(0, 8, L.IO_F64, p(obs), p(done), L.ACT_U8, 0.02, 1, 0, 9, p(obs), z) == L.ERR_UNSUPPORTED  # integrator
    assert lib.emei_terminal_io(0, 8, L.IO_F64, p(obs), 0, z, p(done), z) == L.OK
    torch.cuda.synchronize()
