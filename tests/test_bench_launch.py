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
