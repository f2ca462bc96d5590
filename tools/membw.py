"""HBM bandwidth microbenchmarks with torch ops (fill / copy / read-reduce) to calibrate the roofline."""
import torch, time
def bench(fn, nbytes, name, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name}: {nbytes/ms/1e6:.0f} GB/s ({ms:.3f} ms)")
n = 1 << 29  # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
bench(lambda: a.fill_(1.0), n * 4, "fill (write-only)")
bench(lambda: a.zero_(), n * 4, "zero_ (write-only)")
bench(lambda: b.copy_(a), n * 8, "copy (read+write bytes)")
bench(lambda: a.sum(), n * 4, "sum (read-only)")
bench(lambda: torch.add(a, 1.0, out=b), n * 8, "add out (r+w)")
