"""Wave lifetimes against the launch span for the staged 4-state kernels (a -DEMEI_CLOCK_PROBE build): EMEI_HIP_LIB=$PWD/gpurun_abl_clk.so python tools/pend_span.py"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from emei_amd import _lib
from emei_amd.sharding import ShardedRollout
M64=(1<<64)-1
for env, n, T, fr, tu in (("BoundaryInvertedPendulumSwingUp", 262144, 250, 4, "pend_tu_ip3_f64"), ("BoundaryInvertedPendulumBalancing", 262144, 250, 4, "pend_tu_ip1_f64"),
                          ("BoundaryInvertedPendulumBalancing", 196608, 250, 4, "pend_tu_ip1_f64"), ("CartPoleSwingUp", 65536, 1000, 1, "pend_tu_cp0_f64"), ("CartPoleSwingUp", 131072, 1000, 1, "pend_tu_cp0_f64")):
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] not in env: continue
    sr = ShardedRollout(env, n, T, freq_rate=fr, real_time_scale=0.02)
    sr.make_synthetic_inputs()
    for _ in range(300): sr.run_pass()  # ~0.2 s: the clocks settle (bench.py's settle phase)
    torch.cuda.synchronize()
    ms = sr.timed_launches_ms(50)
    fn = getattr(_lib.lib(), "emei_debug_stats_" + tu)
    out = (C.c_ulonglong * 32)()
    assert fn(out) == 0
    for _ in range(20): sr.run_pass()
    torch.cuda.synchronize()
    assert fn(out) == 0
    end, cyc, ticks, items, nbegin, longest = int(out[27]), int(out[28]), int(out[29]), int(out[30]), int(out[31]), int(out[23])
    span = (end - ((~nbegin) & M64)) / 20.0  # 20 back-to-back launches: first begin to last end
    items //= 20; ticks /= 20; cyc /= 20
    print(f"{env} n={n}: kernel {ms*1e3:.1f} us, waves {items}, mean lifetime {ticks/items/100:.1f} us, longest {longest/100:.1f} us, span {span/100:.1f} us, mean/span {ticks/items/span*100:.1f} %, clock {cyc/ticks*0.1:.3f} GHz", flush=True)
    if "--xcc" in sys.argv:  # a -DEMEI_CLOCK_HIST_XCC build: mean lifetime per XCD
        print("    mean lifetime per XCD (us):", " ".join(f"{int(out[k]) / max(int(out[8 + k]), 1) / 100:.1f}" for k in range(8)),
              " waves per XCD:", " ".join(str(int(out[8 + k]) // 20) for k in range(8)), flush=True)
    else:
        print("    lifetimes in 50 us bins (waves per launch):", " ".join(str(int(out[k]) // 20) for k in range(16)), flush=True)
