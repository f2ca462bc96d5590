"""bench.py --gpus N as the driver types it (no launcher): the parent starts the ranks itself and relays their status.
On this CPU-only container the ranks cannot run (bench.py needs a GPU), which is exactly what the parent must report:
a non-zero exit and no JSON line.  The GPU-side contract is tests/test_gpu_bench_contract.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_relays_the_ranks_failure_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_bench_contract.py")
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "needs a GPU" in p.stderr  # both ranks got as far as bench.py's own check: the launcher did start them


def test_self_launch_refuses_a_profiler_preload():
    """`rocprofv3 ... -- python bench.py --gpus 2` would profile the launcher parent (GPU already initialised
    by the preload) — refused before anything is spawned."""
    e = dict(os.environ, ROCPROFILER_REGISTER_FORCE_LOAD="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "profiler" in p.stderr and not p.stdout.strip()


def test_self_launch_accepts_a_profiler_path_variable_without_a_preload():
    """module files and containers export ROCPROFILER_* / ROCP_* variables (paths, register settings) with no
    profiler attached — not a reason to refuse; and --allow-profiler-env overrides a detected preload.  Without a GPU the ranks
    then fail at bench.py's own check, which proves the launcher was started."""
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_bench_contract.py")
    for extra_env, extra_args in (({"ROCPROFILER_METRICS_PATH": "/opt/rocm/share/rocprofiler-sdk", "ROCP_HSA_INTERCEPT": "0"}, []),
                                  ({"ROCPROFILER_REGISTER_FORCE_LOAD": "1"}, ["--allow-profiler-env"])):
        e = dict(os.environ, **extra_env)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            e.pop(k, None)
        p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"] + extra_args, cwd=ROOT, env=e,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode not in (0, 2) and "needs a GPU" in p.stderr, (extra_env, p.stderr[-500:])


def test_rccl_summary_parses_an_info_log(tmp_path):
    """bench.rccl_summary: best-effort reading of NCCL_DEBUG=INFO output (version, ranks, the lines naming algorithm / protocol)."""
    sys.path.insert(0, ROOT)
    import bench

    log = tmp_path / "rccl.log"
    log.write_text("host:1:1 [0] NCCL INFO RCCL version 2.22.3+hip6.4\n"
                   "host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 8 cudaDev 0 busId 1000 - Init START\n"
                   "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\nhost:1:1 [0] NCCL INFO Channel 15/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "host:1:1 [0] NCCL INFO AllGather: opCount 0 sendbuff 0x1 recvbuff 0x2 count 16777216 datatype 7 -> algo 1 proto 2 time 1.5\n")
    r = bench.rccl_summary(str(log), 8)
    assert r["version"].startswith("2.22") and r["nranks"] == 8 and r["expected_nranks"] == 8 and r["channels"] == 16
    assert len(r["algo_proto_lines"]) == 1 and "algo 1 proto 2" in r["algo_proto_lines"][0]
    assert "error" in bench.rccl_summary(str(tmp_path / "missing.log"), 8)


def test_parent_never_imports_torch_before_launching():
    """the self-launching parent must not initialise the GPU (a later exec / fork from such a process takes the box down):
    its code path ends in self_launch() before the first `import torch`"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("self_launch(a.gpus)") < main.index("import torch")
    body = src[src.index("def self_launch("):src.index("def main():")]
    assert "import torch" not in body and "torch.cuda" not in body


def test_host_cpu_share_is_capped_by_the_cgroup_quota(monkeypatch):
    """bench.host_cpu_share(): the affinity mask capped by cpu.max (cgroup v2) — a one-GPU box shows 256 CPUs and grants 16."""
    import builtins
    import io

    sys.path.insert(0, ROOT)
    import bench

    real_open = builtins.open
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)

    def fake(quota):
        def _open(path, *a, **k):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                return io.StringIO(quota)
            if str(path).startswith("/sys/fs/cgroup/cpu/"):
                raise OSError("no cgroup v1 here")
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert bench.host_cpu_share() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.host_cpu_share() == 256
    monkeypatch.setattr(builtins, "open", fake("150000 100000\n"))  # 1.5 CPUs -> 2 threads
    assert bench.host_cpu_share() == 2


def test_extras_guard_prints_the_line_and_leaves_when_the_deadline_passes(tmp_path):
    """bench.ExtrasGuard (N > 1: the measurements after the timed region run under a deadline): a main thread that never comes back
    from an "extra" still yields rank 0's line, marked, and exit status 0; other ranks leave silently; a guard that is finished in
    time lets the caller print, once."""
    prog = tmp_path / "guard.py"
    prog.write_text(
        "import sys, time, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "rank, mode = int(sys.argv[1]), sys.argv[2]\n"
        "out = {'value': 1.5, 'n_gpus': 2}\n"
        "g = bench.ExtrasGuard(out, rank, 0.3).start()\n"
        "if mode == 'hang':\n"
        "    out['compute_only'] = 7\n"
        "    time.sleep(60)\n"
        "    print('NOT REACHED')\n"
        "else:\n"
        "    assert g.finish() and not g.finish()\n"
        "    time.sleep(0.6)\n"
        "    print(json.dumps(out))\n")
    r0 = subprocess.run([sys.executable, str(prog), "0", "hang"], capture_output=True, text=True, timeout=60)
    assert r0.returncode == 0 and "NOT REACHED" not in r0.stdout
    line = __import__("json").loads(r0.stdout.strip())
    assert line["value"] == 1.5 and line["compute_only"] == 7 and "extras_timed_out" in line
    r1 = subprocess.run([sys.executable, str(prog), "1", "hang"], capture_output=True, text=True, timeout=60)
    assert r1.returncode == 0 and r1.stdout.strip() == ""
    ok = subprocess.run([sys.executable, str(prog), "0", "fine"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and "extras_timed_out" not in ok.stdout and ok.stdout.count("{") == 1
