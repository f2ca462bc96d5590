import os, sys, tempfile
sys.path.insert(0, ".")
os.environ["EMEI_DATASET_PATH"] = tempfile.mkdtemp()
import emei_amd, torch
env = emei_amd.make("CartPoleSwingUp-v0", num_envs=65536)
obs, _ = env.reset(seed=0)
obs, reward, terminal, truncated, info = env.step(torch.ones(65536, dtype=torch.int64, device=obs.device))
obs_T, rew_T, term_T, trunc_T = env.rollout(torch.randint(0, 2, (1000, 65536), device=obs.device, dtype=torch.uint8))
hop = emei_amd.make("HopperRunning-v0", num_envs=4096, integrator="rk4", obs_noise_params=(1e-3, 1e-2),
                    terminate_when_unhealthy=False, auto_reset=True)
from emei_amd import datasets
data, info = datasets.collect(hop, 200, policy=lambda obs: torch.tanh(obs[:, 3:6]))
datasets.save_for_env(hop, data, "my-policy")
d = hop.get_dataset("my-policy")
print("README example ok:", obs_T.shape, d["observations"].shape, info)
