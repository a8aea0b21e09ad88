"""Scratch: env-steps/s of the drop-in API paths (PCIe-inclusive vs device-resident)."""
import time, numpy as np, torch
from dm_control_amd import suite
from dm_control_amd.vec_env import VecEnv
B, T = 8192, 300
env = suite.load('cheetah', 'run', task_kwargs={'random': 0}, environment_kwargs={'batch_size': B, 'device_init': True})
env.reset()
acts = [np.random.uniform(-1, 1, (B, 6)) for _ in range(8)]
for t in range(20): env.step(acts[t % 8])
t0 = time.perf_counter()
for t in range(T): ts = env.step(acts[t % 8])
dt = time.perf_counter() - t0
print('Environment.step numpy (H2D actions, D2H obs/reward/warn, OrderedDict): %.2f ms/step -> %.2f M env-steps/s' % (dt/T*1e3, B*T/dt/1e6))
env.physics.free()
v = VecEnv('cheetah', 'run', B, seed=0)
v.reset()
for t in range(20): v.step(acts[t % 8])
t0 = time.perf_counter()
for t in range(T): v.step(acts[t % 8])
dt = time.perf_counter() - t0
print('VecEnv numpy: %.2f ms/step -> %.2f M env-steps/s' % (dt/T*1e3, B*T/dt/1e6))
v.close()
v = VecEnv('cheetah', 'run', B, seed=0, torch_io=True)
v.reset()
ta = [torch.rand(B, 6, device='cuda')*2 - 1 for _ in range(8)]
for t in range(20): v.step(ta[t % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(T): obs, rew, done, info = v.step(ta[t % 8])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('VecEnv torch (device-resident, clones obs/reward each step): %.3f ms/step -> %.2f M env-steps/s' % (dt/T*1e3, B*T/dt/1e6))
# with a small policy network on the same stream
pol = torch.nn.Sequential(torch.nn.Linear(17, 256), torch.nn.Tanh(), torch.nn.Linear(256, 6), torch.nn.Tanh()).cuda()
with torch.no_grad():
  for t in range(20): obs, rew, done, info = v.step(pol(obs))
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(T): obs, rew, done, info = v.step(pol(obs))
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('VecEnv torch + 17-256-6 MLP policy in the loop: %.3f ms/step -> %.2f M env-steps/s' % (dt/T*1e3, B*T/dt/1e6))
