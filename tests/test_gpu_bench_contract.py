"""bench.py's output contract (the driver parses the ONE JSON line rank 0 prints): required keys, their types, and the
arithmetic that ties them together — `value` to `ms_per_step`, `roofline.achieved` to the kernel duration and the
algorithmic bytes per launch, `frac` to achieved / peak.  Run as the driver runs it, in a child process; the N = 2
launch is rehearsed with the ranks sharing the one GPU over gloo (RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict,
            "cpu_baseline": dict}


def _line(cmd, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout: " + p.stdout[-500:]
    return json.loads(lines[0])


def _check_common(j, n_gpus, steps, warmup):
    for k, t in REQUIRED.items():
        if k == "cpu_baseline" and n_gpus > 1:
            continue  # timed on rank 0 at N = 1 only
        assert k in j, k
        assert isinstance(j[k], t) or (t is float and isinstance(j[k], int)), (k, type(j[k]))
    assert "vs_baseline" in j and j["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert (j["metric"], j["unit"], j["higher_is_better"], j["scaling"]) == ("env-steps/s", "env-steps/s", True, "weak")
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (n_gpus, steps, warmup)
    assert j["dtype"] == "f64" and "synthetic" in j["data"] and "workload" in j["config"] and "model" not in j["config"]
    per_pass = j["config"]["env_steps_per_bench_step"]
    assert per_pass == j["config"]["envs_per_gpu"] * j["config"]["horizon"] * n_gpus
    assert j["value"] == pytest.approx(per_pass / (j["ms_per_step"] * 1e-3), rel=1e-6)
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.0 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 22 * j["config"]["envs_per_gpu"] * j["config"]["horizon"] // (
        j["config"]["horizon"] // j["config"]["chunk_steps"])
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert "pend_rollout_staged_kernel" in r["kernel"]
    assert j["launch"]["ranks"] == n_gpus and len(j["launch"]["devices"]) == n_gpus
    assert r["traffic"] is None or "traffic_source" in r


def test_default_single_gpu_line():
    steps, warmup = 5, 2
    j = _line([sys.executable, "bench.py", "--steps", str(steps), "--warmup", str(warmup)])
    _check_common(j, 1, steps, warmup)
    assert "65 536" in j["config"]["workload"] and "configs[1]" in j["config"]["workload"]
    # the whole timed region is K launches: a pass cannot be faster than its kernel
    assert j["ms_per_step"] >= j["roofline"]["kernel_ms"] * 0.98
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # SURVEY 8d's other two CPU numbers: the same C rollout on one thread, and the reference's own call pattern
    one, ref = c["one_core"], c["reference_style"]
    assert one["cores"] == 1 and 0 < one["value"] <= c["value"] * 1.05 and one["sample"]
    assert ref["cores"] == 1 and 1e3 < ref["value"] < one["value"] and ref["ms_per_1000_steps"] == pytest.approx(1e6 / ref["value"], rel=1e-6)
    r = j["roofline"]
    assert isinstance(r["profile_stale"], bool)
    if r["profile_kernel_avg_us"]:
        assert r["profile_stale"] == (abs(r["kernel_ms"] * 1e3 - r["profile_kernel_avg_us"]) > 0.1 * r["profile_kernel_avg_us"])
    assert j["value"] > 1e7  # north_star's floor


@pytest.mark.parametrize("how", ["direct", "under_launcher"])
def test_two_rank_line_over_gloo_sharing_the_gpu(how):
    """`python bench.py --gpus 2` exactly as the driver types it (bench.py starts its own ranks), and the same line when a
    launcher already did (torch.distributed.run)."""
    steps, warmup = 3, 1
    args = ["bench.py", "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup)]
    if how == "direct":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29633"] + args
    env = {"EMEI_BENCH_SHARE_GPU": "1", "EMEI_BENCH_BACKEND": "gloo"}
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):  # the test itself may run under a launcher
        e.pop(k, None)
    e.update(env)
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]
    j = json.loads(lines[0])
    _check_common(j, 2, steps, warmup)
    cfg = j["config"]
    assert cfg["envs_per_gpu"] == 131072 and cfg["gather"] == "per_chunk" and cfg["chunk_steps"] == 125
    x = j["xgmi"]
    assert x["inbound_bytes_per_rank_per_pass"] == 131072 * 1000 * 16  # one peer's whole [T, n, 4] float32 return
    assert x["peak"] == pytest.approx(7 * 153.6 / 2)  # inbound = half of the links' bidirectional rate
    la = j["launch"]
    assert la["ranks"] == 2 and la["backend"] == "gloo" and len(la["devices"]) == 2 and all(d.startswith("cuda:0") for d in la["devices"])
    assert la["self_launched"] == (how == "direct")
    # beside `value` (the default per_chunk exchange) the kernels' own rate per rank and the final-gather job
    co = j["compute_only"]
    assert len(co["per_rank"]) == 2 and co["value"] == pytest.approx(sum(co["per_rank"])) and co["unit"] == "env-steps/s"
    assert co["per_rank"][0] == pytest.approx(131072 * 125 / (j["roofline"]["kernel_ms"] * 1e-3), rel=1e-6)  # rank 0's own launches
    assert co["value"] >= j["value"]  # the exchange can only cost
    vf = j["value_final_gather"]
    assert vf["gather"] == "final" and vf["value"] == pytest.approx(2 * 131072 * 1000 / (vf["ms_per_step"] * 1e-3), rel=1e-6)
    assert vf["value"] >= 0.9 * j["value"]
    # the other form of the exchange (batched point-to-point transfers) timed in the same run, same gather mode
    vd = j["value_direct_exchange"]
    assert vd["exchange"] == "direct" and vd["gather"] == "per_chunk" and vd["unit"] == "env-steps/s"
    assert vd["value"] == pytest.approx(2 * 131072 * 1000 / (vd["ms_per_step"] * 1e-3), rel=1e-6) and vd["xgmi_inbound_gbs"] > 0
    assert "rccl" not in j  # gloo rehearsal: RCCL's own report is switched on for the nccl backend only


def test_collective_is_the_other_exchange_when_direct_is_the_runs_own():
    steps, warmup = 2, 1
    e = dict(os.environ, EMEI_BENCH_SHARE_GPU="1", EMEI_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--exchange", "direct",
                        "--envs-per-gpu", "16384"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["config"]["exchange"] == "direct" and j["value_collective_exchange"]["exchange"] == "collective"
    assert "value_direct_exchange" not in j


def test_peer_write_exchange_as_the_runs_own_form():
    """--exchange peer_write (opt-in): the rollout kernel writes the observation return into every rank's hipIpc-mapped buffer; the
    line says so, the collective is timed beside it, the launch that ran is the peers kernel's plain twin for the kernel-only timing."""
    e = dict(os.environ, EMEI_BENCH_SHARE_GPU="1", EMEI_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--exchange", "peer_write",
                        "--envs-per-gpu", "16384"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["config"]["exchange"] == "peer_write" and "WITHOUT a collective" in j["config"]["obs_allgather"]
    assert j["value"] > 0 and j["value_collective_exchange"]["exchange"] == "collective" and "extras_error" not in j
    assert j["compute_only"]["value"] > j["value"]


@pytest.mark.parametrize("how", ["hang", "raise"])
def test_a_hang_or_an_error_after_the_timed_region_does_not_cost_the_line(how):
    """The measurements that follow the timed region at N > 1 (compute_only, value_final_gather, the other exchange form) run under a
    deadline (bench.ExtrasGuard): a hang there leaves the headline number in the line, marked; an exception is recorded."""
    e = dict(os.environ, EMEI_BENCH_SHARE_GPU="1", EMEI_BENCH_BACKEND="gloo", EMEI_BENCH_TEST_EXTRAS=how)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "16384", "--extras-timeout", "5"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "xgmi" in j and "compute_only" not in j
    assert ("extras_timed_out" in j) == (how == "hang") and ("extras_error" in j) == (how == "raise")


def test_a_failing_rank_fails_the_self_launched_bench():
    """exit status of the launcher is relayed: an impossible workload size makes every rank raise"""
    e = dict(os.environ, EMEI_BENCH_SHARE_GPU="1", EMEI_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--envs-per-gpu", "-5"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_other_workloads_carry_their_own_cpu_baseline():
    """configs[2] / configs[3]: the CPU oracle of the SAME env kind is timed beside the GPU number (bounded sample)."""
    for wl, frac in (("invpend", "InvertedPendulum"), ("cheetah", "HalfCheetahRunning")):
        j = _line([sys.executable, "bench.py", "--workload", wl, "--steps", "3", "--warmup", "1"], timeout=600)
        c = j["cpu_baseline"]
        assert c["kind"] == "port" and c["value"] > 0 and frac in c["sample"] and c["cores"] >= 1
        assert c["one_core"]["cores"] == 1 and 0 < c["one_core"]["value"] <= c["value"] * 1.05
        assert j["launch"]["ranks"] == 1
