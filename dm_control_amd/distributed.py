"""Multi-GPU use of the batched step: one process per GPU, envs sharded.

Environment instances are independent (the reference's only data parallelism
is one OS process per env, /root/reference/dm_control/scripts/vec_env.py:
334-459), so the env axis is split contiguously across ranks and NO collective
runs inside `step`.  The reporting path gathers per-env episode returns with
one all-gather (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU
node, "gloo" in CPU tests); at 8192 envs that is 32 KB per rank, i.e.
latency-bound, so no bucketing or ring tuning applies.
"""

import os

import numpy as np


def shard_range(total_envs, world_size, rank):
  """Contiguous [start, stop) of the env axis owned by `rank`.

  The first `total % world` ranks own one extra env, so every env has exactly
  one owner and shard sizes differ by at most one.
  """
  if not 0 <= rank < world_size:
    raise ValueError('rank {} outside world of size {}'.format(rank, world_size))
  base, extra = divmod(int(total_envs), int(world_size))
  start = rank*base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def env_seeds(base_seed, total_envs, world_size, rank):
  """Per-env seeds `base_seed + global_index` (vec_env.py:462-465 style)."""
  start, stop = shard_range(total_envs, world_size, rank)
  return np.arange(start, stop, dtype=np.int64) + int(base_seed)


def init_process_group(backend=None, single_rank_group=False):
  """Initialises torch.distributed from the torchrun environment.

  A world of one needs no process group; `single_rank_group=True` creates it
  anyway (bench.py under a one-rank torchrun, so that the RCCL path is
  exercised on a single-GPU box too).
  """
  import torch
  import torch.distributed as dist
  if dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  if world == 1 and not single_rank_group:
    return 0, 1
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29500')
  if backend is None:
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  kwargs = {}
  if backend == 'nccl':
    local = int(os.environ.get('LOCAL_RANK', rank))
    torch.cuda.set_device(local)
    kwargs['device_id'] = torch.device('cuda', local)
  dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
  return rank, world


def gather_episode_returns(local_returns, total_envs=None):
  """All-gathers per-env returns; every rank gets the [total_envs] vector.

  `local_returns` is a 1-D torch tensor (CUDA for nccl, CPU for gloo).  Shards
  may differ in length by one, so tensors are padded to the longest shard for
  the collective and trimmed afterwards.
  """
  import torch
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return local_returns
  world = dist.get_world_size()
  n_local = torch.tensor([local_returns.numel()], device=local_returns.device,
                         dtype=torch.int64)
  sizes = [torch.zeros_like(n_local) for _ in range(world)]
  dist.all_gather(sizes, n_local)
  sizes = [int(s.item()) for s in sizes]
  longest = max(sizes)
  padded = torch.zeros(longest, device=local_returns.device,
                       dtype=local_returns.dtype)
  padded[:local_returns.numel()] = local_returns
  parts = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(parts, padded)
  out = torch.cat([p[:n] for p, n in zip(parts, sizes)])
  if total_envs is not None and out.numel() != total_envs:
    raise RuntimeError('gathered {} returns, expected {}'.format(
        out.numel(), total_envs))
  return out
