import time, torch, ctypes as C, sys
sys.path.insert(0,'.')
from emei_amd.engine import Engine, _ptr, _stream
from emei_amd import _lib as L
N=65536
e=Engine("CartPoleSwingUp",N,max_episode_steps=1000); e.reset(0)
a=torch.randint(0,2,(N,),device=e.device,dtype=torch.uint8)
out=e.alloc_outputs(None)
def bench(f,n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
print("engine.step(out=)      %.2f us"%bench(lambda: e.step(a,auto_reset=True,out=out)))
print("engine.step(alloc)     %.2f us"%bench(lambda: e.step(a,auto_reset=True)))
lib=L.lib(); h=e._h; pa,po,pr,pd=_ptr(a),_ptr(out[0]),_ptr(out[1]),_ptr(out[2]); st=_stream()
print("raw ctypes emei_step   %.2f us"%bench(lambda: lib.emei_step(h,pa,0,po,pr,pd,1,st)))
print("_stream()              %.2f us"%bench(lambda: _stream(),20000))
print("torch empty kernel-ish %.2f us"%bench(lambda: out[1].zero_()))
