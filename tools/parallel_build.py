"""`__graft_entry__.build()` with the per-model hipcc runs spread over the host
cores: the calls `build()` would make are recorded first (build_model patched
to a recorder), then executed in a process pool, then `build()` itself runs
(everything is cached by then, so it only verifies)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one(job):
  from dm_control_amd import build
  args, kwargs = job
  t0 = time.time()
  try:
    out = build.build_model(*args, **kwargs)
    return os.path.basename(out), time.time() - t0, None
  except Exception as e:  # pylint: disable=broad-except
    return None, time.time() - t0, '%r %r: %s' % (args[1:], kwargs, str(e)[-300:])


def main():
  import __graft_entry__ as entry
  from dm_control_amd import build
  os.environ.setdefault('DMC_ALLOW_OVERBUDGET', '0')
  jobs = []
  real = build.build_model

  def record(*args, **kwargs):
    jobs.append((args, kwargs))
    return 'recorded'
  build.build_model = record
  try:
    entry.build()
  finally:
    build.build_model = real
  workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 1)
  t0 = time.time()
  with mp.get_context('fork').Pool(workers) as pool:
    for name, dt, err in pool.imap_unordered(_one, jobs):
      if err:
        print('FAILED', err, flush=True)
  print('%d builds in %.0f s on %d workers' % (len(jobs), time.time() - t0, workers),
        flush=True)
  entry.build()
  print('build() ok', flush=True)


if __name__ == '__main__':
  main()
