"""Minimal stand-in for the third-party `dm_env` package (not installable here).

Only the API surface the reference's hot path touches is provided
(SURVEY.md Appendix C): `TimeStep`, `StepType`, `Environment`, `specs`,
and the `restart/transition/termination/truncation` helpers.  Used at
/root/reference/dm_control/rl/control.py:21-22,88-92,120-123.  If the real
`dm_env` is importable it is preferred, so agents written against it see the
genuine types.
"""

try:  # pragma: no cover - exercised only where dm_env exists
  from dm_env import Environment, StepType, TimeStep, specs  # pylint: disable=unused-import
  from dm_env import restart, termination, transition, truncation  # pylint: disable=unused-import
  HAVE_DM_ENV = True
except ImportError:
  HAVE_DM_ENV = False
  import abc
  import collections
  import enum

  from dm_control_amd._dm_env import specs  # pylint: disable=g-import-not-at-top

  class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self):
      return self is StepType.FIRST

    def mid(self):
      return self is StepType.MID

    def last(self):
      return self is StepType.LAST

  class TimeStep(collections.namedtuple(
      'TimeStep', ['step_type', 'reward', 'discount', 'observation'])):
    __slots__ = ()

    def first(self):
      return self.step_type == StepType.FIRST

    def mid(self):
      return self.step_type == StepType.MID

    def last(self):
      return self.step_type == StepType.LAST

  class Environment(metaclass=abc.ABCMeta):
    """Abstract RL environment (reset/step/specs)."""

    @abc.abstractmethod
    def reset(self):
      pass

    @abc.abstractmethod
    def step(self, action):
      pass

    @abc.abstractmethod
    def observation_spec(self):
      pass

    @abc.abstractmethod
    def action_spec(self):
      pass

    def reward_spec(self):
      return specs.Array(shape=(), dtype=float, name='reward')

    def discount_spec(self):
      return specs.BoundedArray(shape=(), dtype=float, minimum=0., maximum=1.,
                                name='discount')

    def close(self):
      pass

    def __enter__(self):
      return self

    def __exit__(self, *unused):
      self.close()

  def restart(observation):
    return TimeStep(StepType.FIRST, None, None, observation)

  def transition(reward, observation, discount=1.0):
    return TimeStep(StepType.MID, reward, discount, observation)

  def termination(reward, observation):
    return TimeStep(StepType.LAST, reward, 0.0, observation)

  def truncation(reward, observation, discount=1.0):
    return TimeStep(StepType.LAST, reward, discount, observation)
