"""N>1 path on CPU: env sharding + the reporting all-gather (gloo, 2 ranks).

bench.py goes through the same three functions (`shard_range`, `env_seeds`,
`gather_episode_returns`) and `init_process_group`; here they run with the
gloo backend and ragged shards."""

import os
import socket

import numpy as np
import pytest

from dm_control_amd import distributed


def test_shard_ranges_partition_the_env_axis():
  for total in (1, 7, 8192, 8193, 1024):
    for world in (1, 2, 3, 8):
      spans = [distributed.shard_range(total, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == total
      for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0
      sizes = [b - a for a, b in spans]
      assert max(sizes) - min(sizes) <= 1
  assert distributed.shard_range(8192, 8, 3) == (3072, 4096)
  with pytest.raises(ValueError):
    distributed.shard_range(8, 2, 2)
  np.testing.assert_array_equal(distributed.env_seeds(100, 10, 2, 1),
                                np.arange(105, 110))


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, total, queue):
  import torch
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    RANK=str(rank), WORLD_SIZE=str(world))
  r, w = distributed.init_process_group('gloo')
  assert (r, w) == (rank, world)
  # exactly what bench.py --global-batch does per rank: shard, per-env seeds,
  # (simulate), gather the returns of all envs
  start, stop = distributed.shard_range(total, world, rank)
  seeds = distributed.env_seeds(1000, total, world, rank)
  assert len(seeds) == stop - start and seeds[0] == 1000 + start
  # each rank "simulates" its shard: the return of env i is a function of its
  # global seed only, so the gathered vector does not depend on the sharding
  local = torch.from_numpy((seeds - 1000).astype('float32'))*0.5 + 1.0
  full = distributed.gather_episode_returns(local, total)
  queue.put((rank, full.numpy().copy()))
  import torch.distributed as dist
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_all_gather_of_episode_returns_two_ranks():
  import torch.multiprocessing as mp
  ctx = mp.get_context('spawn')
  queue = ctx.Queue()
  total, world, port = 11, 2, _free_port()    # ragged: shards of 6 and 5
  procs = [ctx.Process(target=_worker, args=(r, world, port, total, queue))
           for r in range(world)]
  for p in procs:
    p.start()
  results = dict(queue.get(timeout=100) for _ in range(world))
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  expected = np.arange(total, dtype=np.float32)*0.5 + 1.0
  for rank in range(world):
    np.testing.assert_array_equal(results[rank], expected)


def test_gather_is_identity_without_process_group():
  import torch
  x = torch.arange(4.0)
  assert distributed.gather_episode_returns(x) is x


def _bench(*argv, env=None):
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  e = dict(os.environ if env is None else env)
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    e.pop(k, None)
  return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(argv),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e,
                        timeout=280, universal_newlines=True)


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus N` as the driver calls it (no torchrun): the parent
  starts N ranks itself; --dry-run takes the launcher, the sharding and the
  reporting all-gather with gloo and no kernels (scripts/vec_env.py:396-465 is
  the reference's fan-out)."""
  import json
  for argv, total, shard0 in (
      (['--gpus', '2', '--dry-run'], 16384, [0, 8192]),
      (['--gpus', '3', '--dry-run', '--global-batch', '11'], 11, [0, 4])):
    proc = _bench(*argv)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == 1, proc.stdout            # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line['dry_run'] and line['gather_ok']
    assert line['world_size_seen'] == int(argv[1]) == line['n_gpus']
    assert line['global_batch'] == line['gathered_returns'] == total
    assert line['rank0_shard'] == shard0
    assert line['scaling'] == ('strong' if '--global-batch' in argv else 'weak')


@pytest.mark.timeout(300)
def test_bench_launcher_propagates_a_failing_rank():
  """A rank that dies (here: asked for more ranks than the launcher started)
  must turn into a non-zero exit code of the parent, not a hang."""
  env = dict(os.environ, RANK='0')      # pretend to be a rank of a 1-rank world
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  proc = subprocess.run([sys.executable, os.path.join(root, 'bench.py'),
                         '--gpus', '2', '--dry-run'], env=env, timeout=200,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE)
  assert proc.returncode != 0
