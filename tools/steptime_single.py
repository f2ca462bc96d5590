"""Latency of the gym-style single-env step() (host values in and out) per env family, and of its host pieces for CartPole."""
import sys
import time

sys.path.insert(0, '.')
import emei_amd

for name, kw in (("CartPoleSwingUp-v0", {}), ("BoundaryInvertedPendulumSwingUp-v0", {}), ("HalfCheetahRunning-v0", {}), ("HopperRunning-v0", {})):
    env = emei_amd.make(name, **kw)
    env.reset(seed=0)
    a = 1 if name.startswith("CartPole") else env.action_space.sample()
    for _ in range(50):
        env.step(a)
    best = 1e9
    for rep in range(5):
        t = time.perf_counter()
        for _ in range(1000):
            env.step(a)
        best = min(best, (time.perf_counter() - t) / 1000 * 1e6)
    line = f"{name:40s} single-env step: {best:6.1f} us"
    if name.startswith("CartPole"):
        eng = env.engine
        t = time.perf_counter()
        for _ in range(1000):
            eng.step_host(a)
        line += "   (Engine.step_host alone %.1f us)" % ((time.perf_counter() - t) / 1000 * 1e6)
    print(line)
