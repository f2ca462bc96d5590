"""ShardedRollout.run_pass (bench.py's multi-GPU path) on real device buffers, streams and events: every gather
mode, with 2 ranks sharing the one GPU over gloo (content of every peer block checked) and with a 1-rank RCCL group
(the collective the 8-GPU run uses).  tests/host/sharded_rollout_check.py is the per-rank program."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROG = os.path.join(ROOT, "tests", "host", "sharded_rollout_check.py")


def _run(nproc, port, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), PROG, *args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


@pytest.mark.parametrize("gather,chunk", [("per_chunk", 16), ("per_chunk", 64), ("per_step", 1), ("final", 0)])
def test_two_ranks_share_the_gpu_over_gloo(gather, chunk):
    out = _run(2, 29611, "--backend", "gloo", "--gather", gather, "--chunk", str(max(chunk, 1)),
               "--horizon", "64" if gather != "per_step" else "24")
    assert "world=2" in out and "pend_rollout_staged_kernel" in out or gather == "per_step"


@pytest.mark.parametrize("gather", ["per_chunk", "final"])
def test_one_rank_rccl_group(gather):
    out = _run(1, 29612, "--backend", "nccl", "--gather", gather, "--chunk", "16", "--envs", "65536")
    assert "world=1 backend=nccl" in out


@pytest.mark.parametrize("gather", ["per_chunk", "final"])
def test_direct_exchange(gather):
    """The point-to-point form of the exchange (sharding._allgather_direct): 2 ranks over gloo (peer blocks checked) and the
    1-rank RCCL group (no peer: the local copy alone)."""
    out = _run(2, 29613, "--backend", "gloo", "--gather", gather, "--chunk", "16", "--horizon", "64", "--exchange", "direct")
    assert "world=2" in out
    out = _run(1, 29614, "--backend", "nccl", "--gather", gather, "--chunk", "16", "--envs", "65536", "--exchange", "direct")
    assert "world=1 backend=nccl" in out


@pytest.mark.parametrize("backend,nproc", [("gloo", 2), ("nccl", 1)])
def test_slow_consumer_never_sees_an_overwritten_buffer(backend, nproc):
    """a learner that reads every observation block on a stream of its own (zoo/util.py:54-59 reads every obs)
    while the producer runs collectives ahead.  The receive buffers alternate, so without the released[b] event the gather two
    collectives later would overwrite what the consumer is still about to read; under ObsExchange.reading() every copy the
    consumer takes is the block of its own collective.  2 gloo ranks sharing the GPU, and the 1-rank RCCL group (the asynchronous
    collective path the 8-GPU run uses)."""
    out = _run(nproc, 29615 + nproc, "--backend", backend, "--gather", "per_chunk", "--chunk", "8", "--horizon", "64", "--slow-consumer",
               *(("--envs", "65536") if backend == "nccl" else ()))
    assert f"world={nproc}" in out


@pytest.mark.parametrize("nproc,extra", [(2, ()), (3, ()), (2, ("--slow-consumer",))])
def test_peer_write_exchange(nproc, extra):
    """The observation return by peer writes from the rollout kernel (sharding.PeerWriteExchange; emei_set_obs_peers +
    emei_peer_buffer_*): 2 and 3 processes sharing the one GPU map each other's gathered buffers through hipIpc, every rank's kernel
    stores its rows into every rank's buffer, and every block of every rank is checked against a local recomputation — no collective
    moves an observation (gloo carries the 64-byte handles and the barriers)."""
    out = _run(nproc, 29620 + 2 * nproc + len(extra), "--backend", "gloo", "--gather", "per_chunk", "--chunk", "16", "--horizon", "64",
               "--exchange", "peer_write", *extra)
    assert f"world={nproc}" in out and "peers" in out
