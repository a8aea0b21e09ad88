import os, sys, subprocess
for flags in ['', '-DDMC_ABLATE_OBS', '-DDMC_ABLATE_SOLVER', '-DDMC_ABLATE_SOLVER -DDMC_ABLATE_CONTACT', '-DDMC_ABLATE_SOLVER -DDMC_ABLATE_CONTACT -DDMC_ABLATE_OBS']:
  env = dict(os.environ, DMC_EXTRA_FLAGS=flags)
  out = subprocess.run([sys.executable, 'gpu_perf_probe.py', 'cheetah', 'run', '8192'], env=env, capture_output=True, text=True)
  print('FLAGS [%s]' % flags); print(out.stdout[-400:], out.stderr[-300:])
