"""Copies gpurun_out/final/* (tools/final_measurements.sh) into profiles/ and prints a table."""
import json
import os
import shutil
O = 'gpurun_out/final'


def last_json(path):
  return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])


for tag in ('cheetah_run_b8192', 'humanoid_walk_b8192'):
  f = last_json('%s/pmc_FETCH_SIZE_%s.json' % (O, tag))['FETCH_SIZE']
  w = last_json('%s/pmc_WRITE_SIZE_%s.json' % (O, tag))['WRITE_SIZE']
  p = 'profiles/r01_pmc_%s_f32.json' % tag
  out = json.load(open(p))
  out.update(FETCH_SIZE_KB=f['median'], WRITE_SIZE_KB=w['median'],
             traffic_bytes_per_launch=(f['median'] + w['median'])*1024,
             kernel_stats_same_round=last_json('%s/stats_%s.json' % (O, tag)))
  json.dump(out, open(p, 'w'), indent=1)
  shutil.copy('%s/stats_%s/s_kernel_stats.csv' % (O, tag),
              'profiles/r01_final_%s_f32_kernel_stats.csv' % tag)
  open('profiles/r01_final_bench_under_rocprof_%s.json' % tag, 'w').write(
      json.dumps(last_json('%s/bench_under_rocprof_%s.log' % (O, tag))) + '\n')
for fn in sorted(os.listdir(O)):
  if fn.startswith('bench_') and fn.endswith('.json') and 'rocprof' not in fn:
    d = last_json(os.path.join(O, fn))
    tag = fn[6:-5]
    if os.path.exists('profiles/r01_pmc_%s_f32.json' % tag):
      d['roofline']['traffic'] = json.load(
          open('profiles/r01_pmc_%s_f32.json' % tag))['traffic_bytes_per_launch']
    open('profiles/r01_final_' + fn, 'w').write(json.dumps(d) + '\n')
    e = d['cpu_baseline']['qpos_rel_err']
    print('%-26s %10.4g env-steps/s  kernel %.4f ms  cpu %9.3g  rel-err median %.1e max %.1e' % (
        tag, d['value'], d['roofline']['kernel_ms_avg'], d['cpu_baseline']['value'],
        e['median'], e['max']))
