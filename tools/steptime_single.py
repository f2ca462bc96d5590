import time, numpy as np, torch, sys
sys.path.insert(0, '.')
import emei_amd
for name, kw in (("CartPoleSwingUp-v0", {}), ("HalfCheetahRunning-v0", {}), ("HopperRunning-v0", {})):
    env = emei_amd.make(name, **kw)
    env.reset(seed=0)
    a = env.action_space.sample()
    for _ in range(20): env.step(a)
    t = time.perf_counter()
    for _ in range(300): env.step(a)
    print(name, "single-env step: %.1f us" % ((time.perf_counter() - t) / 300 * 1e6))
