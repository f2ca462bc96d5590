"""How many host threads the C oracle should use on a GPU box: prints the cgroup CPU quota, the affinity mask, and the fused
InvertedPendulum rollout's rate at a few OMP_NUM_THREADS (each in a child process: libgomp reads the variable at load)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
rng = np.random.default_rng(1)
n = 262144
st = rng.standard_normal((n, 4)) * 5e-3
a = rng.uniform(-3, 3, (25, n)).astype(np.float32)
r = O.body_rollout("ip", "boundary_swingup", st, a, 4, 0.02)
t = time.perf_counter(); k = 0
while time.perf_counter() - t < 4.0:
    r = O.body_rollout("ip", "boundary_swingup", r["state"], a, 4, 0.02, reuse=r); k += 1
print("%%.3g env-steps/s" %% (n * 25 * k / (time.perf_counter() - t)))
""" % ROOT

if __name__ == "__main__":
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(p):
            print(p, open(p).read().strip())
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
    for th in sys.argv[1:] or ["8", "16", "32", "64", "256"]:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, OMP_NUM_THREADS=th), capture_output=True, text=True)
        print("OMP_NUM_THREADS", th, out.stdout.strip(), out.stderr.strip()[-300:])
