"""Hopper domain (cf. /root/reference/dm_control/suite/hopper.py)."""

from dm_control_amd import codegen
from dm_control_amd import engine
from dm_control_amd.rl import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.utils import containers

SUITE = containers.TaggedTasks()

_CONTROL_TIMESTEP = .02
_DEFAULT_TIME_LIMIT = 20
_STAND_HEIGHT = 0.6     # fused into the kernel (csrc/dmc_kernels.hip, TASK_HOPPER)
_HOP_SPEED = 2


def get_model_and_assets():
  return common.read_model('hopper.xml'), common.ASSETS


def _make(hopping, time_limit, random, environment_kwargs):
  phys_kw, task_kw, env_kw = common.split_kwargs(environment_kwargs)
  physics = Physics.from_xml_string(*get_model_and_assets(), **phys_kw)
  task = Hopper(hopping=hopping, random=random, **task_kw)
  physics.set_task_params(iparam=1 if hopping else 0)
  return control.Environment(physics, task, time_limit=time_limit,
                             control_timestep=_CONTROL_TIMESTEP, **env_kw)


@SUITE.add('benchmarking')
def stand(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  """Returns a Hopper that strives to stand upright (hopper.py:52-60)."""
  return _make(False, time_limit, random, environment_kwargs)


@SUITE.add('benchmarking')
def hop(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  """Returns a Hopper that strives to hop forward (hopper.py:63-71)."""
  return _make(True, time_limit, random, environment_kwargs)


class Physics(engine.Physics):
  """Physics with the Hopper helpers (hopper.py:74-90).

  Fused observation layout: [qpos[1:], qvel, log1p(touch_toe, touch_heel)].
  """

  _TASK = codegen.TASK_HOPPER
  # one env per wavefront up to 2048 envs, two up to 4096 (measured cross-overs)
  _COOP_POLICY = ((1024, 128), (2048, 64), (4096, 32))
  _COOP_POLICY_F64 = ((1024, 128), (8192, 32))

  def height(self):
    """Height of the torso's centre of mass above the foot's."""
    xipos = self.named.data.xipos
    return xipos['torso', 'z'] - xipos['foot', 'z']

  def speed(self):
    return self.named.data.sensordata['torso_subtreelinvel'][..., 0]

  def touch(self):
    return self.fused_observation()[..., -2:]


class Hopper(base.Task):
  """Standing / hopping task (hopper.py:93-140)."""

  def __init__(self, hopping, random=None, device_init=False):
    self._hopping = hopping
    super().__init__(random=random, device_init=device_init)

  def initialize_episode(self, physics):
    if self._device_init:
      physics.batch.init_episode(self.device_seed())
    else:
      qpos = randomizers.randomized_qpos(self, physics)
      physics.data.qpos[:] = qpos[0] if physics.batch_size is None else qpos
    self._timeout_progress = 0
    super().initialize_episode(physics)

  def get_observation(self, physics):
    m = physics.model
    return self._obs_dict(physics, [('position', m.nq - 1, False),
                                    ('velocity', m.nv, False),
                                    ('touch', 2, False)])
