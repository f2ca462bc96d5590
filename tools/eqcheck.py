import torch, sys
sys.path.insert(0,'.')
from emei_amd.engine import Engine
for prec in ("ref","f32"):
  for adt in (torch.uint8, torch.int32, torch.int64):
    for (N,T) in ((1024,37),(4096,64),(128,16),(64,100)):
        a=Engine("CartPoleSwingUp",N,freq_rate=2,precision=prec,max_episode_steps=20,seed=5)
        b=Engine("CartPoleSwingUp",N,freq_rate=2,precision=prec,max_episode_steps=20,seed=5)
        a.reset(5); b.reset(5)
        acts=torch.randint(0,2,(T,N),device=a.device).to(adt)
        obs,rew,done=a.rollout(acts,auto_reset=True)
        ok=True
        for t in range(T):
            o,r,d=b.step(acts[t],auto_reset=True)
            ok &= torch.equal(o,obs[t]) and torch.equal(r,rew[t]) and torch.equal(d,done[t])
        ok &= torch.equal(a.get_state(),b.get_state())
        ok &= torch.equal(a.compact_done(), b.compact_done())
        print(prec,adt,N,T,'OK' if ok else 'MISMATCH', int((done!=0).sum()))
