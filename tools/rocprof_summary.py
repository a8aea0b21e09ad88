"""Summarises rocprofv3 CSV output for one kernel (profiles/README.md).

  python tools/rocprof_summary.py stats <dir> [kernel]    # --kernel-trace --stats run
  python tools/rocprof_summary.py pmc <dir> [kernel]      # --pmc run: median per launch

Walks <dir> for *_kernel_stats.csv / *_counter_collection.csv.
"""
import csv
import json
import os
import statistics
import sys


def _find(root, suffix):
  out = []
  for d, _, files in os.walk(root):
    out += [os.path.join(d, f) for f in files if f.endswith(suffix)]
  return sorted(out)


def stats(root, kernel):
  for path in _find(root, 'kernel_stats.csv'):
    with open(path) as f:
      for row in csv.DictReader(f):
        if row.get('Name', '').startswith(kernel):
          calls = int(row['Calls'])
          total, mx = float(row['TotalDurationNs']), float(row['MaxNs'])
          print(json.dumps(dict(
              file=os.path.basename(path), kernel=row['Name'], calls=calls,
              total_ms=total/1e6, avg_ms=total/calls/1e6, max_ms=mx/1e6,
              avg_ms_without_longest=(total - mx)/max(1, calls - 1)/1e6)))


def pmc(root, kernel):
  values = {}
  for path in _find(root, 'counter_collection.csv'):
    with open(path) as f:
      for row in csv.DictReader(f):
        if row.get('Kernel_Name', '').startswith(kernel):
          values.setdefault(row['Counter_Name'], []).append(
              float(row['Counter_Value']))
  print(json.dumps({k: dict(median=statistics.median(v), n=len(v),
                            max=max(v)) for k, v in values.items()}))


if __name__ == '__main__':
  mode, root = sys.argv[1], sys.argv[2]
  kernel = sys.argv[3] if len(sys.argv) > 3 else 'dmc_step'
  {'stats': stats, 'pmc': pmc}[mode](root, kernel)
