"""Long random-action rollouts through VecEnv (episode resets included) on the kernel
shapes Physics picks by batch size: warning bits, finiteness, reward range, constraint
statistics."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dm_control_amd import vec_env
for domain, task, nenv, steps in (('humanoid', 'walk', 8192, 990), ('humanoid', 'run', 1024, 1990), ('cheetah', 'run', 512, 1990), ('walker', 'walk', 1000, 1500), ('hopper', 'hop', 777, 1500)):
  env = vec_env.VecEnv(domain, task, nenv, seed=11, torch_io=True)
  ph = env.environment.physics
  env.reset()
  gen = torch.Generator(device='cuda').manual_seed(1)
  nu = ph.model.nu
  t0 = time.time()
  tot = torch.zeros(nenv, device='cuda')
  for t in range(steps):
    a = torch.rand(nenv, nu, device='cuda', generator=gen)*2 - 1
    obs, rew, done, info = env.step(a)
    tot += rew
  torch.cuda.synchronize()
  import dm_control_amd.wrapper as W
  warn = ph.batch.read(W.FIELD_WARN)
  st = ph.batch.read(W.FIELD_STATS)
  print('%s-%s %d envs x %d steps: %s | warn bits set in %d envs | finite %s | reward range %.3f..%.3f | ncon max %d nefc max %d iters max %d | %.1f s' % (
      domain, task, nenv, steps, ph.kernel_shape[:40], int((warn != 0).sum()), bool(torch.isfinite(tot).all()), float(rew.min()), float(rew.max()), st[0].max(), st[1].max(), st[2].max(), time.time() - t0), flush=True)
  env.close()
