"""Scratch perf probe: per-launch time and solver iteration statistics."""
import os
import sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import suite, wrapper
name, task, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else 'f32'
env = suite.load(name, task, task_kwargs={'random': 1}, environment_kwargs={'batch_size': B, 'device_init': True, 'precision': prec})
p = env.physics; b = p.batch
env.reset()
nsub = env._n_sub_steps
rs = np.random.RandomState(0)
acts = [rs.uniform(-1, 1, (B, p.model.nu)) for _ in range(8)]
its, ncons, nefcs = [], [], []
for t in range(60):
  p.set_control(acts[t % 8]); p.step(nsub, check=False)
  st = b.read(wrapper.FIELD_STATS)
  its.append(st[2].copy()); ncons.append(st[0].copy()); nefcs.append(st[1].copy())
its = np.array(its); nefcs = np.array(nefcs)
print('iters mean %.2f p99 %d max %d | per-wave max mean %.2f | nefc mean %.1f max %d' % (
    its.mean(), np.percentile(its, 99), its.max(), its.reshape(60, -1, 64).max(axis=2).mean(), nefcs.mean(), nefcs.max()))
print('iters histogram', np.bincount(its.ravel().astype(int))[:30])
print('per-step max iters', its.max(axis=1)[:20], 'nefc at max', [int(nefcs[t, its[t].argmax()]) for t in range(20)])
b.sync(); b.timer_start()
for t in range(100):
  p.set_control(acts[t % 8]); p.step(nsub, check=False)
ms, n = b.timer_stop()
print('%s-%s B=%d %s: %.3f ms/launch (host ctrl copies included) -> %.2f M env-steps/s' % (name, task, B, prec, ms/n, B/(ms/n)/1e3))
print('warn', int((b.read(wrapper.FIELD_WARN) != 0).sum()))
