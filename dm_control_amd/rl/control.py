"""Batched `control.Environment` for the MI355X physics step.

Same orchestration, step counting, time-limit and `TimeStep` conventions as the
reference (/root/reference/dm_control/rl/control.py:28-161), applied to B
instances that advance in lock step:

  * `reset()`  -> FIRST, reward None, discount None              (:77-92)
  * `step(a)`  -> MID with discount 1.0, or LAST at the time limit with
                  discount 1.0, or LAST with `task.get_termination`  (:94-123)
  * the call after LAST resets and ignores the action             (:97-98)
  * `n_sub_steps` from `control_timestep` via `compute_n_steps`   (:164-190)

The `n_sub_steps` physics steps, the task observation and the reward of one
control step are a single kernel launch (`physics.step(n_sub_steps)`).
With an unbatched Physics the TimeStep holds python floats and un-batched
arrays exactly like the reference; with a batch every leaf gains a leading B.
"""

import collections

import numpy as np

from dm_control_amd import _dm_env as dm_env

specs = dm_env.specs

FLAT_OBSERVATION_KEY = 'observations'


class Environment(dm_env.Environment):
  """Physics-based RL environment over a batch of instances."""

  def __init__(self, physics, task, time_limit=float('inf'),
               control_timestep=None, n_sub_steps=None,
               flat_observation=False):
    self._physics, self._task = physics, task
    self._flat_observation = bool(flat_observation)
    dt = physics.timestep()
    if control_timestep is None:
      substeps = 1 if n_sub_steps is None else n_sub_steps
    elif n_sub_steps is None:
      substeps = compute_n_steps(control_timestep, dt)
    else:
      raise ValueError('pass either control_timestep or n_sub_steps, not both '
                       '(got {} and {})'.format(control_timestep, n_sub_steps))
    self._n_sub_steps = substeps
    # a float on purpose (compared with >=): time_limit / (dt * n_sub), the
    # expression of rl/control.py:72-73; inf / x stays inf
    self._step_limit = time_limit / (dt * substeps)
    self._step_count = 0
    self._reset_next_step = True

  def _observe(self):
    observation = self._task.get_observation(self._physics)
    return (flatten_observation(observation) if self._flat_observation
            else observation)

  def reset(self):
    """Starts a new episode (all instances) and returns the first TimeStep."""
    self._reset_next_step, self._step_count = False, 0
    with self._physics.reset_context():
      self._task.initialize_episode(self._physics)
    return dm_env.TimeStep(dm_env.StepType.FIRST, None, None, self._observe())

  def step(self, action):
    """Advances every instance by one control step (one kernel launch)."""
    if self._reset_next_step:        # the call after LAST ignores the action
      return self.reset()
    task, physics = self._task, self._physics
    task.before_step(action, physics)
    physics.step(self._n_sub_steps)
    task.after_step(physics)
    reward, observation = task.get_reward(physics), self._observe()
    self._step_count += 1
    batch = getattr(physics, 'batch_size', None)
    # time limit => LAST with discount 1; otherwise the task may terminate
    discount = (1.0 if self._step_count >= self._step_limit
                else task.get_termination(physics))
    if discount is None:
      ones = 1.0 if batch is None else np.ones(batch, np.float64)
      return dm_env.TimeStep(dm_env.StepType.MID, reward, ones, observation)
    self._reset_next_step = True
    if batch is not None and np.isscalar(discount):
      discount = np.full(batch, discount, np.float64)
    return dm_env.TimeStep(dm_env.StepType.LAST, reward, discount, observation)

  def action_spec(self):
    return self._task.action_spec(self._physics)

  def step_spec(self):
    return self._task.step_spec(self._physics)

  def observation_spec(self):
    try:
      return self._task.observation_spec(self._physics)
    except NotImplementedError:
      observation = self._task.get_observation(self._physics)
      if self._flat_observation:
        observation = flatten_observation(observation)
      return _spec_from_observation(observation)

  @property
  def step_count(self):
    """Control steps taken in the current episode (the counter behind the
    time limit, rl/control.py:110-114); assignable when a checkpoint is restored."""
    return self._step_count

  @step_count.setter
  def step_count(self, value):
    self._step_count = int(value)
    self._reset_next_step = self._step_count >= self._step_limit

  @property
  def physics(self):
    return self._physics

  @property
  def task(self):
    return self._task

  @property
  def batch_size(self):
    return getattr(self._physics, 'batch_size', None)

  def control_timestep(self):
    return self.physics.timestep() * self._n_sub_steps


def compute_n_steps(control_timestep, physics_timestep, tolerance=1e-8):
  """Physics steps per control step (the contract of rl/control.py:164-190):
  the ratio must be a whole number >= 1 to within `tolerance`."""
  ratio = control_timestep / physics_timestep
  whole = int(round(ratio))
  if control_timestep < physics_timestep:
    raise ValueError('a control step of {} s is shorter than one physics step '
                     '({} s)'.format(control_timestep, physics_timestep))
  if abs(ratio - whole) > tolerance:
    raise ValueError('a control step of {} s is {} physics steps of {} s; it '
                     'has to be a whole number of them'.format(
                         control_timestep, ratio, physics_timestep))
  return whole


def _spec_from_observation(observation):
  result = collections.OrderedDict()
  for key, value in observation.items():
    value = np.asarray(value)
    result[key] = specs.Array(value.shape, value.dtype, name=key)
  return result


# `control.Physics`, `control.Task`, `control.PhysicsError` live in rl/abstract.py
from dm_control_amd.rl.abstract import Physics, PhysicsError, Task  # noqa: E402,F401  pylint: disable=g-import-not-at-top


def flatten_observation(observation, output_key=FLAT_OBSERVATION_KEY):
  """Concatenates observation leaves in key order (control.py:368-393).

  Leaves keep a leading batch axis if they have one: an unbatched leaf is
  ravelled to 1-D, a batched [B, ...] leaf to [B, -1].
  """
  if not isinstance(observation, collections.abc.MutableMapping):
    raise ValueError('flatten_observation takes a dict of arrays, got a {}'
                     .format(type(observation).__name__))
  ordered = isinstance(observation, collections.OrderedDict)
  names = list(observation) if ordered else sorted(observation)
  leaves = [np.asarray(observation[name]) for name in names]
  batch = getattr(observation, 'batch_size', None)
  if batch is None:
    arrays = [leaf.ravel() for leaf in leaves]
    flat = np.concatenate(arrays)
  else:
    flat = np.concatenate([leaf.reshape(batch, -1) for leaf in leaves], axis=1)
  out = type(observation)([(output_key, flat)])
  if batch is not None:
    out.batch_size = batch
  return out


class BatchedObservation(collections.OrderedDict):
  """OrderedDict of [B, ...] leaves that remembers its batch size."""

  batch_size = None
