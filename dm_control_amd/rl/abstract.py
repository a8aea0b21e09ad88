"""Abstract `Physics` / `Task` interfaces and `PhysicsError`.

The contract `control.Environment` relies on (cf. the ABCs at
/root/reference/dm_control/rl/control.py:202-365); re-exported from
`dm_control_amd.rl.control` under the reference's names.
"""

import abc
import contextlib


class Physics(metaclass=abc.ABCMeta):
  """Simulates a physical environment (control.py:202-261)."""

  @abc.abstractmethod
  def step(self, n_sub_steps=1):
    """Updates the simulation state `n_sub_steps` times."""

  @abc.abstractmethod
  def time(self):
    """Elapsed simulation time in seconds."""

  @abc.abstractmethod
  def timestep(self):
    """Simulation timestep."""

  def set_control(self, control):
    raise NotImplementedError('set_control is not supported.')

  @contextlib.contextmanager
  def reset_context(self):
    """Resets on entry, runs `after_reset` on exit (control.py:226-247)."""
    try:
      self.reset()
    except PhysicsError:
      pass
    yield self
    self.after_reset()

  @abc.abstractmethod
  def reset(self):
    """Resets internal variables of the physics simulation."""

  @abc.abstractmethod
  def after_reset(self):
    """Runs after resetting internal variables of the physics simulation."""

  def check_divergence(self):
    """Raises a `PhysicsError` if the simulation state is divergent."""


class PhysicsError(RuntimeError):
  """Raised if the state of the physics simulation becomes divergent."""


class Task(metaclass=abc.ABCMeta):
  """Defines a task in a `control.Environment` (control.py:268-365)."""

  @abc.abstractmethod
  def initialize_episode(self, physics):
    """Sets the state of the environment at the start of each episode."""

  @abc.abstractmethod
  def before_step(self, action, physics):
    """Updates the task from the provided action."""

  def after_step(self, physics):
    """Optional hook after the physics step."""

  @abc.abstractmethod
  def action_spec(self, physics):
    """Specification of valid actions."""

  def step_spec(self, physics):
    raise NotImplementedError()

  @abc.abstractmethod
  def get_observation(self, physics):
    """Returns an observation from the environment."""

  @abc.abstractmethod
  def get_reward(self, physics):
    """Returns a reward from the environment."""

  def get_termination(self, physics):
    """If the episode should end, returns a final discount, otherwise None."""

  def observation_spec(self, physics):
    raise NotImplementedError()
