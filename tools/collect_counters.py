"""Assembles profiles/counters_<code object>_b<batch>.json from the rocprofv3
`--pmc` passes of one bench workload (tools/r02_measure.sh), keyed by the code
object the passes ran on -- bench.py only reports counter figures whose code
object matches the one it is running (profiles/README.md).

  python tools/collect_counters.py <gpurun_out/measure dir> <tag> <bench json line file>

Unit corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and
WRITE_SIZE are in KB per launch as rocprofv3 reports them; on gfx950 FETCH_SIZE
counts wide (16-byte per lane) coalesced streaming reads at half their bytes --
the step kernel's loads are 4-byte per lane, coalesced over the wave, for which
the counter is exact, so no correction is applied (stated in the output).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(directory):
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rocprof_summary.py'),
                        'pmc', directory], stdout=subprocess.PIPE, universal_newlines=True)
  try:
    return json.loads(out.stdout.strip().splitlines()[-1])
  except (ValueError, IndexError):
    return {}


def main():
  base, tag, bench_file = sys.argv[1:4]
  with open(bench_file) as f:
    line = json.loads([l for l in f if l.startswith('{')][-1])
  code_object = line['config']['code_object']
  batch = line['config']['global_batch']//line['n_gpus']
  counters = {}
  for name in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU'):
    counters.update(pmc(os.path.join(base, 'pmc_%s_%s' % (name, tag))))
  med = {k: v['median'] for k, v in counters.items()}
  out = {
      'code_object': code_object, 'batch': batch, 'workload': line['config']['workload'],
      'source': 'rocprofv3 --pmc, separate passes, medians per dmc_step launch '
                '(tools/r02_measure.sh)',
      'counters': med}
  if 'FETCH_SIZE' in med and 'WRITE_SIZE' in med:
    out['traffic_bytes_per_launch'] = (med['FETCH_SIZE'] + med['WRITE_SIZE'])*1024
    out['traffic_note'] = ('FETCH_SIZE + WRITE_SIZE (KB) x 1024; 4-byte-per-lane '
                           'coalesced accesses: no gfx950 wide-read correction applies')
  if 'SQ_INSTS_VALU' in med:
    out['valu_insts_per_launch'] = med['SQ_INSTS_VALU']
  if 'SQ_WAVES' in med:
    out['waves_per_launch'] = med['SQ_WAVES']
  if 'SQ_ACTIVE_INST_VALU' in med and 'SQ_WAVE_CYCLES' in med and med['SQ_WAVE_CYCLES']:
    out['valu_busy'] = med['SQ_ACTIVE_INST_VALU']/med['SQ_WAVE_CYCLES']
  tagname = code_object.replace('dmc_', '').replace('.hsaco', '')
  path = os.path.join(ROOT, 'gpurun_out', 'measure', 'counters_%s_b%d.json' % (tagname, batch))
  with open(path, 'w') as f:
    json.dump(out, f, indent=1)
  print(path, json.dumps(out))


if __name__ == '__main__':
  main()
