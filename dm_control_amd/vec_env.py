"""Gym-style vector environment over the batched suite environments.

Calling convention of the reference's only data-parallel mechanism,
`SubprocVecEnv` (/root/reference/dm_control/scripts/vec_env.py:127-135,
346-352, 447-472): `reset() -> obs[B, D]`; `step(actions[B, nu]) ->
(obs[B, D], rewards[B], dones[B], infos)`; an env that finishes is reset
inside `step` and its last observation is reported as
`info['terminal_observation']`.  Here the B "workers" are lanes of one kernel
launch instead of OS processes, so stepping needs no pipes or pickling.

Two I/O modes:
  * numpy (default): actions are host arrays, results are host arrays; every
    step costs one H2D and a few D2H copies (PCIe-inclusive path).
  * torch (`torch_io=True`): actions are CUDA tensors read in place, results
    are CUDA tensors aliasing the kernel's output buffers, everything runs on
    torch's current stream -- no host round trip in the loop.
"""

import numpy as np

from dm_control_amd import suite
from dm_control_amd import wrapper
from dm_control_amd.rl import control


class VecEnv:
  """B synchronous suite environments with auto-reset."""

  def __init__(self, domain_name, task_name, num_envs, seed=None,
               device=0, precision='f32', torch_io=False, task_kwargs=None,
               environment_kwargs=None):
    env_kw = dict(environment_kwargs or {})
    env_kw.update(batch_size=int(num_envs), device=device,
                  precision=precision, flat_observation=True)
    env_kw.setdefault('device_init', bool(torch_io))
    task_kw = dict(task_kwargs or {})
    task_kw.setdefault('random', seed)
    self._env = suite.load(domain_name, task_name, task_kwargs=task_kw,
                           environment_kwargs=env_kw)
    self.num_envs = int(num_envs)
    self._physics = self._env.physics
    self._batch = self._physics.batch
    self._torch = bool(torch_io)
    self._nsub = self._env._n_sub_steps          # pylint: disable=protected-access
    self._step_limit = self._env._step_limit     # pylint: disable=protected-access
    self._count = 0
    spec = self._env.action_spec()
    self.action_low, self.action_high = spec.minimum, spec.maximum
    self.action_dim = int(spec.shape[0])
    self.observation_dim = int(self._batch.model.info.nobs)
    if self._torch:
      from dm_control_amd import torch_io as tio
      tio.use_current_stream(self._batch)
      self._obs_t = tio.field_tensor(self._batch, wrapper.FIELD_OBS)
      self._rew_t = tio.field_tensor(self._batch, wrapper.FIELD_REWARD)

  @property
  def environment(self):
    return self._env

  # -- numpy mode -------------------------------------------------------------
  def _obs(self, timestep):
    return timestep.observation[control.FLAT_OBSERVATION_KEY]

  def reset(self):
    """Starts new episodes everywhere; returns obs [B, D]."""
    if self._torch:
      return self._reset_torch()
    self._count = 0
    return self._obs(self._env.reset())

  def step(self, actions):
    """-> (obs [B, D], rewards [B], dones [B], infos list of dicts)."""
    if self._torch:
      return self._step_torch(actions)
    ts = self._env.step(actions)
    infos = [{} for _ in range(self.num_envs)]
    dones = np.full(self.num_envs, ts.last())
    rewards = np.asarray(ts.reward, np.float64)
    obs = self._obs(ts)
    if ts.last():
      for i in range(self.num_envs):
        infos[i]['terminal_observation'] = obs[i].copy()
      obs = self._obs(self._env.reset())          # vec_env.py:346-352
    return obs, rewards, dones, infos

  # -- torch mode ---------------------------------------------------------------
  def _reset_torch(self):
    physics = self._physics
    with physics.reset_context():
      self._env.task.initialize_episode(physics)
    self._count = 0
    return self._obs_t.clone()

  def _step_torch(self, actions):
    import torch
    if not actions.is_cuda:
      raise ValueError('torch_io=True expects CUDA action tensors')
    a = actions.to(self._obs_t.dtype)
    if a.shape != (self.num_envs, self.action_dim):
      raise ValueError('actions must have shape (%d, %d)'
                       % (self.num_envs, self.action_dim))
    if a.stride(1) != 1:
      a = a.contiguous()
    self._keepalive = a
    physics = self._physics
    physics.set_control_device(a.data_ptr(), a.stride(1), a.stride(0))
    physics.step(self._nsub, check=False)
    self._count += 1
    done = self._count >= self._step_limit
    rewards = self._rew_t.clone()
    dones = torch.full((self.num_envs,), bool(done), device=rewards.device)
    infos = {}
    if done:
      infos['terminal_observation'] = self._obs_t.clone()
      obs = self._reset_torch()
    else:
      obs = self._obs_t.clone()
    return obs, rewards, dones, infos

  def close(self):
    self._physics.free()
