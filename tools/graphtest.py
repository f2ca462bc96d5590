import time, torch, sys
sys.path.insert(0,'.')
from emei_amd.engine import Engine
N=65536; K=64
e=Engine("CartPoleSwingUp",N,max_episode_steps=1000); e.reset(0)
acts=torch.randint(0,2,(K,N),device=e.device,dtype=torch.uint8)
outs=[e.alloc_outputs(None) for _ in range(K)]
# eager
for k in range(K): e.step(acts[k],auto_reset=True,out=outs[k])
torch.cuda.synchronize()
t=time.perf_counter()
for r in range(20):
    for k in range(K): e.step(acts[k],auto_reset=True,out=outs[k])
torch.cuda.synchronize(); eager=(time.perf_counter()-t)/(20*K)*1e6
# graph
s=torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for k in range(3): e.step(acts[k],auto_reset=True,out=outs[k])
torch.cuda.current_stream().wait_stream(s)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(K): e.step(acts[k],auto_reset=True,out=outs[k])
torch.cuda.synchronize()
# check equality of graph replay vs eager from same state
e2=Engine("CartPoleSwingUp",N,max_episode_steps=1000); e2.reset(0)
e.reset(0); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
ref=[e2.step(acts[k],auto_reset=True) for k in range(K)]
ok=all(torch.equal(a[0],b[0]) and torch.equal(a[2],b[2]) for a,b in zip(outs,ref))
for _ in range(3): g.replay()
torch.cuda.synchronize(); t=time.perf_counter()
for r in range(20): g.replay()
torch.cuda.synchronize(); graph=(time.perf_counter()-t)/(20*K)*1e6
print(f"eager {eager:.2f} us/step  graph {graph:.2f} us/step  equal={ok}  -> {N/graph*1e6:.3e} env-steps/s through per-step launches")
