#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused env-step hot path on N GPUs of one node.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  One process per GPU: when N > 1 and the process was
NOT started by a launcher (no WORLD_SIZE in the environment), it starts the N ranks itself — as child processes of
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` —
without touching the GPU, relays rank 0's ONE JSON line and exits with the launcher's status.  Started under
`torch.distributed.run` already, it is a rank (RCCL = backend "nccl").  Rank 0 prints ONE JSON line.

A bench "step" = one pass of the hot path over one batch of synthetic input: `emei_rollout` of
`--horizon` (default 1000 = max_episode_steps of CartPoleSwingUp-v0, register_env.py:19-23) env-steps
for every env of the shard, with device-side auto-reset.
value = env-steps/s over all ranks = n_gpus * envs_per_gpu * horizon * K / time.
Inputs (state, actions) are resident in HBM before the timed region; outputs land in HBM.

N = 1 is BASELINE configs[1] (65 536 CartPoleSwingUp envs, one launch per horizon).  N > 1 shards 131 072
envs per GPU, so that N = 8 IS BASELINE configs[4] (1 048 576 envs), and exchanges the batched observation
return over RCCL: by default (`--gather per_chunk`) the whole [T, n, 4] observation block of a pass is
all-gathered in chunks of `--chunk` steps on a dedicated stream under the following launches; the collective's
inbound volume and rate are reported beside the HBM roofline.  `--gather final` exchanges only the last
observation of a pass, `--gather per_step` one [n, 4] block per env-step (one emei_step launch each).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# 7 point-to-point xGMI links per GPU x ~153 GB/s per link (the figure this task states; it is the link's BIDIRECTIONAL rate):
# what a rank can RECEIVE in a direct all-gather is half of it per link, 7 x 76.8 = 537.6 GB/s
XGMI_LINKS, XGMI_LINK_GBS_BIDIR = 7, 153.6
XGMI_PEAK_GBS = XGMI_LINKS * XGMI_LINK_GBS_BIDIR / 2
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs, max shader clock (MI355X_MICROARCH.md, chip-level parameters)
MEASURED_NS_PER_F64_INST = 2.0  # tools/op_cost.hip at 4 waves per SIMD, every SIMD busy (profiles/r03_op_cost.txt)
MULTI_GPU_SHARD = 131072  # BASELINE configs[4]: 1 048 576 envs / 8 GPUs

WORKLOADS = {
    # name: (env, envs_per_gpu, freq_rate, dt, horizon, act bytes, obs_dim, act_dim)
    "cartpole_swingup": dict(env="CartPoleSwingUp", n=65536, freq_rate=1, dt=0.02, horizon=1000,
                             desc="CartPoleSwingUp-v0, 65 536 parallel envs, freq_ratio=1 (BASELINE configs[1])"),
    "cartpole_balancing": dict(env="CartPoleBalancing", n=65536, freq_rate=1, dt=0.02, horizon=500,
                               desc="CartPoleBalancing-v0, 65 536 parallel envs"),
    "invpend": dict(env="BoundaryInvertedPendulumSwingUp", n=262144, freq_rate=4, dt=0.02, horizon=250,
                    desc="InvertedPendulum forward-Euler, 262 144 parallel envs, freq_ratio=4 (BASELINE configs[2])"),
    "invpend_balancing": dict(env="BoundaryInvertedPendulumBalancing", n=262144, freq_rate=4, dt=0.02, horizon=250,
                              desc="InvertedPendulum Balancing forward-Euler, 262 144 parallel envs, freq_ratio=4 (SURVEY 8d config 3, second variant)"),
    "dpend": dict(env="BoundaryInvertedDoublePendulumSwingUp", n=262144, freq_rate=4, dt=0.02, horizon=100,
                  desc="InvertedDoublePendulum forward-Euler, 262 144 parallel envs, freq_ratio=4 (SURVEY 8f rank 3)"),
    "cheetah": dict(env="HalfCheetahRunning", n=131072, freq_rate=4, dt=0.002, horizon=100,
                    desc="HalfCheetah-style body forward-Euler, 131 072 parallel envs (BASELINE configs[3])"),
    "hopper": dict(env="HopperRunning", n=131072, freq_rate=4, dt=0.002, horizon=100, integrator="rk4",
                   desc="Hopper, RK4 (the reference's default, hopper.py:20-22), 131 072 parallel envs (SURVEY 8f rank 4)"),
}


def algorithmic_bytes_per_env_step(obs_dim, act_bytes):
    """Fused rollout, state in registers: read action, write obs f32 + reward f32 + done u8."""
    return act_bytes + 4 * obs_dim + 4 + 1


def host_cpu_share():
    """Threads worth starting on this host: the affinity mask capped by the cgroup's CPU quota.  A one-GPU box shows 256
    logical CPUs but grants 16 CPUs of time (cpu.max 1600000 100000): 256 OpenMP threads on that quota are throttled to a third
    of what 16 threads reach (profiles/r03_cpu_threads.txt)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                cores = min(cores, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(env, n, freq_rate, dt, integrator="euler", solver="newton", budget_s=12.0):
    """The CPU oracle ("port": the C restatement of the reference's step arithmetic — CartPole pinned bit-exact to the
    reference by tests/golden, the MuJoCo-backed bodies a restatement of MuJoCo's published algorithm) timed on this
    box's host cores with OpenMP, on the same env kind, env count, freq_rate, dt, integrator and solver as the GPU line."""
    import numpy as np

    from oracle import oracle as O

    cores = host_cpu_share()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library loads
    rng = np.random.default_rng(1)
    if env.startswith("CartPole"):
        # the CPU twin of the timed call: a fused multi-step rollout with auto-reset and TimeLimit in C (blocks of envs over the
        # OpenMP threads, no Python between steps), writing the same float32 obs / reward / uint8 done per env-step into
        # buffers that stay allocated — bit-identical to the per-step oracle (tests/test_oracle_golden.py)
        variant = "swingup" if env == "CartPoleSwingUp" else "balancing"
        max_steps = 1000 if variant == "swingup" else 500
        st = O.cartpole_init_state_host(variant, 0, n)
        Tc = 200
        acts = rng.integers(2, size=(Tc, n)).astype(np.uint8)
        r = O.cartpole_rollout_autoreset(variant, st, acts, 0, None, max_steps, freq_rate, dt)  # warm: build, page in the outputs
        t0 = time.perf_counter()
        calls = 0
        while True:
            r = O.cartpole_rollout_autoreset(variant, r["state"], acts, 0, None, max_steps, freq_rate, dt, r["steps"], r["episode"], reuse=r)
            calls += 1
            el = time.perf_counter() - t0
            if el > budget_s:
                break
        out = {"value": n * Tc * calls / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
               "sample": f"{calls} fused rollouts of {Tc} steps x {n} {env} envs (freq_rate {freq_rate}, dt {dt}, auto-reset, TimeLimit {max_steps}, "
                         f"float32 obs / reward and uint8 done written per env-step) with the C oracle (float64, OpenMP x{cores}), {el:.1f} s"}
        # SURVEY 8d's other two CPU numbers.  one_core: the same C rollout on ONE thread (a 2 s sample of a 1/16 shard)
        O.set_threads(1)
        n1 = max(64, n // 16)
        r1 = O.cartpole_rollout_autoreset(variant, st[:n1], acts[:, :n1], 0, None, max_steps, freq_rate, dt)
        t0, calls1 = time.perf_counter(), 0
        while True:
            r1 = O.cartpole_rollout_autoreset(variant, r1["state"], acts[:, :n1], 0, None, max_steps, freq_rate, dt, r1["steps"], r1["episode"], reuse=r1)
            calls1 += 1
            el1 = time.perf_counter() - t0
            if el1 > 2.0:
                break
        O.set_threads(cores)
        out["one_core"] = {"value": n1 * Tc * calls1 / el1, "unit": "env-steps/s", "cores": 1,
                           "sample": f"{calls1} fused rollouts of {Tc} steps x {n1} envs, the same C oracle on one thread, {el1:.1f} s"}
        # reference_style: ONE env stepped the way the reference steps it — a CPython call per step, scalar math.sin / cos, float32
        # np.array derivative, np.append, batched reward / terminal on a [1, 4] row (base_control.py:61-83; BASELINE configs[0]'s
        # call pattern).  Build-authored (oracle/oracle.py:cartpole_reference_style_loop), bit-identical to the C oracle.
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            O.cartpole_reference_style_loop(variant, 1000, 0, freq_rate, dt)
            reps += 1
        el2 = time.perf_counter() - t0
        out["reference_style"] = {"value": 1000 * reps / el2, "unit": "env-steps/s", "cores": 1, "ms_per_1000_steps": el2 / reps * 1e3,
                                  "sample": f"{reps} x 1000 steps of ONE {env} env in a per-step NumPy / math loop that mirrors base_control.py:61-83 "
                                            "(the reference itself took 60.6 ms per 1000 steps in the build container, BASELINE.md)"}
        return out
    # the MuJoCo-backed bodies: the same kind of twin — Tc env-steps per C call (blocks of envs over the OpenMP threads, no Python
    # between steps), float32 obs / reward and uint8 terminal written per env-step, no reset (as the per-step reference) —
    # step for step the per-step oracle's arithmetic (tests/test_oracle_golden.py)
    opt = O.opts(integrator, solver=solver)
    if "InvertedDoublePendulum" in env:
        kind, variant = "dp", [k for k in O.DP_VARIANTS if k.replace("_", "") in env.lower().replace("inverteddoublependulum", "")][0]
        st, lo, nu, Tc = rng.standard_normal((n, 6)) * 5e-3, 1.0, (), 25
    elif "InvertedPendulum" in env:
        kind, variant = "ip", [k for k in O.IP_VARIANTS if k.replace("_", "") in env.lower().replace("invertedpendulum", "")][0]
        st, lo, nu, Tc = rng.standard_normal((n, 4)) * 5e-3, 3.0, (), 25
    elif env == "HalfCheetahRunning":
        kind, variant = "cheetah", None
        st, lo, nu, Tc = rng.standard_normal((n, 18)) * 0.1, 1.0, (6,), 5
    elif env == "HopperRunning":
        kind, variant = "hopper", None
        st, lo, nu, Tc = rng.standard_normal((n, 12)) * 5e-3, 1.0, (3,), 5
        st[:, 1] += 1.25
    else:
        return None
    acts = rng.uniform(-lo, lo, (Tc, n) + nu).astype(np.float32)
    r = O.body_rollout(kind, variant, st, acts[:1], freq_rate, dt, opt)  # warm: build the library, first touch of the state
    r = O.body_rollout(kind, variant, r["state"], acts, freq_rate, dt, opt)  # page in the outputs
    t0 = time.perf_counter()
    calls = 0
    while True:
        r = O.body_rollout(kind, variant, r["state"], acts, freq_rate, dt, opt, reuse=r)
        calls += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    out = {"value": n * Tc * calls / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"{calls} fused rollouts of {Tc} steps x {n} {env} envs (freq_rate {freq_rate}, dt {dt}, {integrator}"
                     + ("" if "Pendulum" in env else f", {solver} solver")
                     + f", no reset, float32 obs / reward and uint8 terminal written per env-step) with the C oracle (float64, OpenMP x{cores}), {el:.1f} s"}
    # one_core: the same C rollout on one thread, a 2 s sample of a 1/16 shard
    O.set_threads(1)
    n1 = max(64, n // 16)
    r1 = O.body_rollout(kind, variant, st[:n1], acts[:, :n1], freq_rate, dt, opt)
    t0, calls1 = time.perf_counter(), 0
    while True:
        r1 = O.body_rollout(kind, variant, r1["state"], acts[:, :n1], freq_rate, dt, opt, reuse=r1)
        calls1 += 1
        el1 = time.perf_counter() - t0
        if el1 > 2.0:
            break
    O.set_threads(cores)
    out["one_core"] = {"value": n1 * Tc * calls1 / el1, "unit": "env-steps/s", "cores": 1,
                       "sample": f"{calls1} fused rollouts of {Tc} steps x {n1} envs, the same C oracle on one thread, {el1:.1f} s"}
    return out


def rccl_summary(path, world):
    """What RCCL reported about itself (NCCL_DEBUG=INFO of rank 0, best effort: the wording differs between versions): version,
    ranks of the communicator, channels, and the lines that name an algorithm / protocol."""
    import re

    info = {"log": path, "nranks": None, "version": None, "channels": None, "algo_proto_lines": []}
    try:
        txt = open(path, errors="replace").read()
    except OSError as e:
        info["error"] = str(e)
        return info
    m = re.search(r"(?:RCCL|NCCL) version ([^\s]+)", txt)
    info["version"] = m.group(1) if m else None
    m = re.search(r"nranks (\d+)", txt)
    info["nranks"] = int(m.group(1)) if m else None
    ch = re.findall(r"(\d+) coll channels|Channel (\d+)/", txt)
    if ch:
        info["channels"] = max(int(a or b) for a, b in ch) + (0 if ch[0][0] else 1)
    seen = []
    for line in txt.splitlines():
        if re.search(r"\b(algo|Algo|protocol|Proto|proto)\b", line) and ("AllGather" in line or "all_gather" in line.lower() or "Algo" in line):
            short = re.sub(r"^.*?NCCL INFO ", "", line)[:160]
            if short not in seen:
                seen.append(short)
        if len(seen) >= 8:
            break
    info["algo_proto_lines"] = seen
    info["expected_nranks"] = world
    return info


class ExtrasGuard:
    """Deadline for the measurements that FOLLOW the timed region at N > 1.  They run collectives of their own (the other exchange form has
    never run on a real multi-GPU node); if one of them hangs, every rank's timer fires at about the same moment (they were started after a
    barrier): rank 0 prints the line it already has, marked `extras_timed_out`, and each rank leaves with os._exit — no collective can be
    waited for any more.  An exception in an extra is recorded under `extras_error` instead of losing the line."""

    def __init__(self, out, rank, seconds):
        import threading

        self.out, self.rank, self.seconds = out, rank, seconds
        self.lock, self.printed = threading.Lock(), False
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True

    def start(self):
        self.timer.start()
        return self

    def _fire(self):
        with self.lock:
            if self.printed:
                return
            self.printed = True
            if self.rank == 0:
                line = dict(self.out)
                line["extras_timed_out"] = f"the measurements after the timed region did not finish within {self.seconds:.0f} s; those present are complete"
                print(json.dumps(line), flush=True)
        os._exit(0)

    def finish(self):
        """True if the caller should print the line (the timer has not fired)."""
        self.timer.cancel()
        with self.lock:
            if self.printed:
                return False
            self.printed = True
            return True


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as children of torch.distributed.run.
    This process never imports torch and never touches a GPU; it relays rank 0's JSON line and the launcher's status."""
    import socket
    import subprocess

    # Under a profiler the preloaded tool library has already initialised the GPU in THIS process, and the ranks would be
    # its grandchildren: counters attach to the wrong process and a GPU-initialised parent must not spawn a launcher on
    # this pool.  Profile a rank directly: `torchrun ... ` started first, rocprofv3 in front of each rank's `python bench.py`.
    # Detected by the preload itself (LD_PRELOAD naming a rocprofiler library, or the variables rocprofv3 sets to force its tool
    # in: ROCP_TOOL_LIBRARIES / ROCPROFILER_REGISTER_FORCE_LOAD) — module files and containers export other ROCPROFILER_* / ROCP_*
    # variables (paths, register settings) with no profiler attached.  --allow-profiler-env overrides.
    preload = os.environ.get("LD_PRELOAD", "")
    attached = "rocprof" in preload or any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
    if attached and "--allow-profiler-env" not in sys.argv:
        sys.stderr.write("bench.py: --gpus N>1 without a launcher refuses to run under a profiler preload "
                         "(rocprofv3 must wrap each rank's process, not this launcher parent); --allow-profiler-env overrides\n")
        raise SystemExit(2)
    with socket.socket() as sk:  # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, EMEI_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)  # stderr passes through
    lines = []
    for line in p.stdout:
        if line.startswith("{"):
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = p.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    if rc == 0:
        print(lines[0], flush=True)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cartpole_swingup", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help="default: the workload's single-GPU size; 131 072 for the CartPole workloads when --gpus > 1")
    ap.add_argument("--gather", default=None, choices=["final", "per_chunk", "per_step"],
                    help="what the observation return exchanges across ranks (default: per_chunk when --gpus > 1, final otherwise)")
    ap.add_argument("--chunk", type=int, default=125, help="steps per launch / per all-gather with --gather per_chunk")
    ap.add_argument("--exchange", default="collective", choices=["collective", "direct", "peer_write"],
                    help="how the observation return crosses ranks: the backend's all-gather, 1-hop transfers to and from every "
                         "peer at once (the xGMI mesh has a link per peer), or no collective at all: the rollout kernel writes every "
                         "observation row into hipIpc-mapped buffers of every rank (CartPole workloads, --gather per_chunk; "
                         "opt-in: it has never run on a multi-GPU node)")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--precision", default="ref", choices=["ref", "f32"])
    ap.add_argument("--integrator", default=None, choices=["euler", "semi_implicit_euler", "rk4"],
                    help="MuJoCo-backed bodies only (mujoco_env.py:70-79); default: the workload's own")
    ap.add_argument("--solver", default="newton", choices=["newton", "sweep1"],
                    help="constraint solver of the HalfCheetah / Hopper workloads: MuJoCo's formulation solved to convergence "
                         "(default) or round 1's single Gauss-Seidel sweep")
    ap.add_argument("--rollout-chunk-steps", type=int, default=0,
                    help="body rollouts as (64 envs) x (k steps) work items: 0 = automatic (the default), -1 = one-piece launches, "
                         "k > 0 = k steps per item (emei_config.rollout_chunk_steps); results do not depend on it")
    ap.add_argument("--settle-ms", type=float, default=60.0, help="untimed clock-settle phase before the warm-up passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-profiler-env", action="store_true", help="self-launch N > 1 ranks even though a profiler preload is detected")
    ap.add_argument("--extras-timeout", type=float, default=300.0,
                    help="N > 1: seconds the additional measurements after the timed region (compute_only, value_final_gather, the other exchange form) may take "
                         "before rank 0 prints the line without them and every rank exits (a hang there must not cost the run its headline number)")
    ap.add_argument("--no-rccl-info", action="store_true", help="do not switch on NCCL_DEBUG=INFO (N > 1, RCCL: the line reports what RCCL chose)")
    ap.add_argument("--per-step-api", action="store_true", help="also time one launch per env-step (emei_step)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)  # does not return

    import numpy as np
    import torch

    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if os.environ.get("EMEI_BENCH_SHARE_GPU"):  # rehearsal of the N>1 path on a one-GPU box (gloo, ranks share cuda:0)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_log = None
    backend = "none (1 rank)"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EMEI_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl" and not a.no_rccl_info and "NCCL_DEBUG" not in os.environ:
            # what RCCL chose (ranks, channels, algorithm / protocol of the all-gather) goes into the line: its INFO log, per rank, to a file
            rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"emei_bench_rccl_{os.getpid()}_r{rank}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,ENV,TUNING", NCCL_DEBUG_FILE=rccl_log)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if not os.path.exists(os.path.join(ROOT, "emei_amd", "libemei_hip.so")):
        if rank == 0:
            ge.build()
        if dist:
            dist.barrier()

    from emei_amd.engine import Engine
    from emei_amd.sharding import ShardedRollout

    w = dict(WORKLOADS[a.workload])
    sharded_cartpole = world > 1 and a.workload.startswith("cartpole")
    N = a.envs_per_gpu or (MULTI_GPU_SHARD if sharded_cartpole else w["n"])
    T = a.horizon or w["horizon"]
    env = w["env"]
    gather = a.gather or ("per_chunk" if world > 1 else "final")
    chunk = a.chunk if T % a.chunk == 0 else T
    sr = ShardedRollout(env, N, T, freq_rate=w["freq_rate"], real_time_scale=w["dt"], precision=a.precision,
                        rank=rank, world=world, device=local_rank, seed=0,
                        integrator=a.integrator or w.get("integrator", "euler"), gather=gather, chunk=chunk, solver=a.solver,
                        exchange_algo=a.exchange, rollout_chunk_steps=a.rollout_chunk_steps)
    desc = w["desc"]
    if sharded_cartpole and a.workload == "cartpole_swingup" and N == MULTI_GPU_SHARD:
        desc = (f"CartPoleSwingUp-v0, {world * N} parallel envs sharded {world}xMI355X ({N} per GPU) with RCCL all-gather of obs"
                + (" (BASELINE configs[4])" if world == 8 else " (BASELINE configs[4]'s shard size)"))
    sr.make_synthetic_inputs()

    # Untimed settle phase before the W warm-up passes: after idling, the first ~10-30 ms of back-to-back
    # launches run ~15 % slower (clocks ramp up; measured: K = 10 gives 0.35 ms per pass after 2 warm-up
    # passes, 0.295 ms after 100), whatever W the caller chose.
    if a.settle_ms > 0:
        sr.run_pass()
        torch.cuda.synchronize()
        t_one = time.perf_counter()
        sr.run_pass()
        torch.cuda.synchronize()
        t_one = time.perf_counter() - t_one
        if dist:  # every rank must run the same number of passes (each one ends in a collective)
            tt = torch.tensor([t_one], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_one = float(tt.item())
        for _ in range(min(2000, int(a.settle_ms * 1e-3 / max(t_one, 1e-6)) + 1)):
            sr.run_pass()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        sr.run_pass()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # HIP events on the stream the kernels are launched on (torch's current stream)
    for _ in range(a.steps):
        sr.run_pass()
    sr.wait_gathers()  # the last passes' observation all-gathers (overlapped with the rollouts) are inside the timed region
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # average launch duration of the dominant kernel over the timed region: at one GPU a pass is
    # exactly one rollout launch, so the event pair around the K passes / K is the kernel's duration
    # (plus the ~1.5 us dependent-launch gap); with several ranks the all-gather sits between launches
    single_launch = world == 1 and sr.n_chunks == 1
    kernel_ms = ev0.elapsed_time(ev1) / a.steps if single_launch else sr.timed_launches_ms(a.steps)
    total_env_steps = world * N * T * a.steps
    value = total_env_steps / el
    bpes = algorithmic_bytes_per_env_step(sr.obs_dim, sr.action_bytes)
    T_launch = sr.chunk  # env-steps per env of ONE launch (= the horizon unless the pass is chunked)
    achieved = bpes * N * T_launch / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch from the rocprofv3 PMC passes (profiles/traffic.json, written from
    # tools/collect_profiles.sh output with the gfx950 FETCH_SIZE correction); only valid for the
    # default shape it was collected on
    traffic = valu = None
    prof = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and N == w["n"] and T_launch == w["horizon"] and a.precision == "ref" and not a.integrator and a.solver == "newton":
        try:
            prof = json.load(open(tpath)).get(a.workload, {})
            traffic, valu = prof.get("bytes_per_launch"), prof.get("valu_insts_per_launch")
        except Exception:
            prof = {}
    # the copied counters describe the kernel the profile was taken on: flag the line when this run's kernel no longer
    # runs as long as that one did (> 10 %: a kernel changed without tools/profile_all.sh, or a very different box)
    prof_us = prof.get("rocprof_kernel_avg_us")
    profile_stale = bool(prof_us) and abs(kernel_ms * 1e3 - prof_us) > 0.10 * prof_us

    out = {
        "metric": "env-steps/s", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if a.precision == "ref" else "f32", "data": "synthetic",
        "config": {"workload": desc, "env": env, "envs_per_gpu": N, "horizon": T, "freq_rate": w["freq_rate"],
                   "real_time_scale": w["dt"], "settle_ms_before_warmup": a.settle_ms, "integrator": a.integrator or w.get("integrator", "euler"),
                   "constraint_solver": a.solver if a.workload in ("cheetah", "hopper") else "n/a (at most one constraint row)", "api": f"emei_rollout ({'one launch per horizon' if sr.n_chunks == 1 else f'{sr.n_chunks} launches of {sr.chunk} steps per horizon'}, device auto-reset)",
                   "env_steps_per_bench_step": world * N * T, "action_dtype": sr.action_dtype_name,
                   "gather": gather, "chunk_steps": sr.chunk, "exchange": a.exchange if world > 1 else "n/a (1 GPU)",
                   "obs_allgather": ({"final": "last [n,obs_dim] observation of each pass",
                                      "per_chunk": f"the whole [T,n,obs_dim] observation return, one all-gather per {sr.chunk}-step chunk",
                                      "per_step": "the [n,obs_dim] observation of every env-step, one all-gather per step"}[gather]
                                     + (" over RCCL, on a dedicated stream under the following launches" if a.exchange != "peer_write" else
                                        " — here WITHOUT a collective: written by the rollout kernel into every rank's hipIpc-mapped buffer, one host barrier per chunk")) if world > 1 else "n/a (1 GPU)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (f"{prof.get('profile', 'profiles/')} via profiles/traffic.json: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE "
                                        "passes of this command, collected separately, NOT measured in this run") if traffic else None,
                     "profile_stale": profile_stale, "profile_kernel_avg_us": prof_us,
                     "kernel": sr.kernel_name, "kernel_ms": kernel_ms, "algorithmic_bytes_per_env_step": bpes,
                     "algorithmic_bytes_per_launch": bpes * N * T_launch},
    }
    if valu:
        # second roofline for the VALU-bound bodies: executed vector instructions per launch (SQ_INSTS_VALU, counted per
        # wave) x 4 cycles of issue each (MI355X_MICROARCH.md: one wave issues a VALU instruction per >= 4 cycles; f64 FMA
        # and f32 alike) / (SIMDs x max clock x kernel time) = the fraction of the chip's vector-issue slots in use
        # `frac_at_measured_rate`: the same count priced at what this chip's SIMDs actually sustain — 2.0 ns per float64 wave-
        # instruction per SIMD with 4 resident waves (FMA / mul / add / cvt / cmp alike; in-kernel clock 2.1 GHz; v_rcp_f64 7.1 ns,
        # float32 / integer 1.2 ns: tools/op_cost.hip, profiles/r03_op_cost.txt)
        out["roofline_valu"] = {"bound": "valu_f64", "valu_insts_per_launch": valu, "cycles_per_inst": 4, "simds": SIMDS,
                                "clock_ghz": CLOCK_GHZ, "frac": valu * 4 / (SIMDS * CLOCK_GHZ * 1e9 * kernel_ms * 1e-3),
                                "measured_ns_per_f64_inst": MEASURED_NS_PER_F64_INST,
                                "frac_at_measured_rate": valu * MEASURED_NS_PER_F64_INST / (SIMDS * kernel_ms * 1e6),
                                "source": f"{prof.get('profile', 'profiles/')} SQ pass via profiles/traffic.json, NOT measured in this run"}
        # `frac_at_probed_clock`: the same count at the shader clock the kernel's waves were MEASURED to run at (tools/clock_probe.py,
        # profiles/r04_clock_probe.txt: s_memtime against the 100 MHz s_memrealtime over every wave's lifetime).  The float64-dense
        # four-waves-per-SIMD kernels are power-limited far below 2.4 GHz (config 3: 1.65 GHz, and 100 % of one instruction per four
        # cycles at that clock); the one-wave-per-SIMD bodies run at 2.38 GHz.
        if prof.get("shader_clock_ghz"):
            out["roofline_valu"]["probed_clock_ghz"] = prof["shader_clock_ghz"]
            out["roofline_valu"]["frac_at_probed_clock"] = valu * 4 / (SIMDS * prof["shader_clock_ghz"] * 1e9 * kernel_ms * 1e-3)
    # what actually ran: the ranks the process group formed, its backend, every rank's device
    me = f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"
    devices = [me]
    if dist:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    out["launch"] = {"ranks": dist.get_world_size() if dist else 1, "backend": backend + (" (RCCL)" if backend == "nccl" else ""),
                     "devices": devices, "self_launched": bool(os.environ.get("EMEI_BENCH_SELF_LAUNCHED"))}
    if world > 1:
        gb = sr.gathered_bytes_per_pass
        out["xgmi"] = {"bound": "xgmi", "inbound_bytes_per_rank_per_pass": gb, "achieved": gb * a.steps / el / 1e9,
                       "peak": XGMI_PEAK_GBS, "unit": "GB/s", "frac": gb * a.steps / el / 1e9 / XGMI_PEAK_GBS,
                       "collectives_per_pass": sr.n_chunks if gather != "final" else 1,
                       "peak_source": f"{XGMI_LINKS} links x {XGMI_LINK_GBS_BIDIR} GB/s bidirectional / 2 (inbound)",
                       "note": "whole-pass average: the all-gathers overlap the rollout launches"
                               + ("; with --gather per_chunk / per_step every rank receives the whole observation return of its peers, "
                                  "so N > 1 is bound by xGMI by design, not by the rollout kernel" if gather != "final" else "")}
    guard = None
    if world > 1:
        # everything below runs collectives of its own AFTER the timed region: under a deadline, so that a hang costs the extras, not the line
        guard = ExtrasGuard(out, rank, a.extras_timeout).start()
        try:
            if os.environ.get("EMEI_BENCH_TEST_EXTRAS") == "hang":  # test hooks (tests/test_gpu_bench_contract.py): a collective that never returns ...
                time.sleep(1e6)
            if os.environ.get("EMEI_BENCH_TEST_EXTRAS") == "raise":  # ... and one that fails
                raise RuntimeError("EMEI_BENCH_TEST_EXTRAS=raise")
            # What the rollout kernels alone sustain, per rank and summed: N envs x the steps of one launch / that launch's duration,
            # measured on every rank with no collective between the launches (timed_launches_ms).  Beside `value` it separates how
            # the KERNEL scales with N from what the chosen observation exchange costs.
            mine = N * T_launch / (kernel_ms * 1e-3)
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
            out["compute_only"] = {"value": float(sum(per_rank)), "unit": "env-steps/s", "per_rank": [float(v) for v in per_rank],
                                   "what": "sum over ranks of envs_per_gpu x steps per launch / rollout-kernel launch duration (HIP events "
                                           "around back-to-back launches, no collective in the span)"}
            if gather != "final":
                # ... and the same job with only the LAST observation of each pass exchanged (--gather final), timed in this run under
                # the same barrier / max-over-ranks rule: `value` keeps the default mode's number
                sf = ShardedRollout(env, N, T, freq_rate=w["freq_rate"], real_time_scale=w["dt"], precision=a.precision, rank=rank,
                                    world=world, device=local_rank, seed=0, integrator=a.integrator or w.get("integrator", "euler"),
                                    gather="final", solver=a.solver)
                sf.make_synthetic_inputs()
                for _ in range(max(a.warmup, 2)):
                    sf.run_pass()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tf = time.perf_counter()
                for _ in range(a.steps):
                    sf.run_pass()
                sf.wait_gathers()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                tf = torch.tensor([time.perf_counter() - tf], dtype=torch.float64, device="cuda")
                dist.all_reduce(tf, op=dist.ReduceOp.MAX)
                out["value_final_gather"] = {"value": total_env_steps / float(tf.item()), "unit": "env-steps/s",
                                             "ms_per_step": float(tf.item()) / a.steps * 1e3, "gather": "final",
                                             "what": "the same job, same run, exchanging only the last [n, obs_dim] observation of each pass"}
                sf.engine.close()
                del sf
            # ... and the same job with the OTHER form of the exchange (`--exchange direct`: 1-hop point-to-point transfers to and from
            # every peer, the shape of the xGMI mesh; or the collective when direct is the run's own), same gather mode, same barrier /
            # max-over-ranks rule: one 8-GPU run answers which of the two the node prefers
            other = "direct" if a.exchange == "collective" else "collective"
            so = ShardedRollout(env, N, T, freq_rate=w["freq_rate"], real_time_scale=w["dt"], precision=a.precision, rank=rank,
                                world=world, device=local_rank, seed=0, integrator=a.integrator or w.get("integrator", "euler"),
                                gather=gather, chunk=chunk, solver=a.solver, exchange_algo=other, rollout_chunk_steps=a.rollout_chunk_steps)
            so.make_synthetic_inputs()
            for _ in range(max(a.warmup, 2)):
                so.run_pass()
            so.wait_gathers()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            to = time.perf_counter()
            for _ in range(a.steps):
                so.run_pass()
            so.wait_gathers()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            to = torch.tensor([time.perf_counter() - to], dtype=torch.float64, device="cuda")
            dist.all_reduce(to, op=dist.ReduceOp.MAX)
            gbo = so.gathered_bytes_per_pass
            out["value_direct_exchange" if other == "direct" else "value_collective_exchange"] = {
                "value": total_env_steps / float(to.item()), "unit": "env-steps/s", "ms_per_step": float(to.item()) / a.steps * 1e3,
                "exchange": other, "gather": gather, "xgmi_inbound_gbs": gbo * a.steps / float(to.item()) / 1e9,
                "what": f"the same job, same run, same gather mode, the observation return as {'batched point-to-point transfers (sharding._allgather_direct)' if other == 'direct' else 'the backend all-gather'}"}
            so.engine.close()
            del so
            if rccl_log:
                out["rccl"] = rccl_summary(rccl_log, world)
        except Exception as e:  # recorded, not fatal: the headline number is already measured
            out["extras_error"] = repr(e)[:500]
    if a.per_step_api and rank == 0:
        out["per_step_api"] = sr.time_per_step_api()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(env, N, w["freq_rate"], w["dt"], a.integrator or w.get("integrator", "euler"), a.solver)
    if guard is not None and not guard.finish():
        time.sleep(30)  # the deadline fired: its thread prints rank 0's line and ends the process
        return
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 and a.exchange == "peer_write":
        sr.close()  # peer-mapped buffers: every rank unmaps before the owners free them
    if dist:
        if "extras_error" in out:  # ranks may be out of step after an error: do not wait for a clean teardown for ever
            import threading

            t = threading.Timer(60.0, os._exit, (0,))
            t.daemon = True
            t.start()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
