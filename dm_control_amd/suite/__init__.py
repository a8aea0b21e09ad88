"""Control Suite loader for the batched MI355X physics step.

Same contract as /root/reference/dm_control/suite/__init__.py:78-150:
`load(domain_name, task_name, task_kwargs, environment_kwargs,
visualize_reward)` returns a `control.Environment`; unknown names raise
ValueError; the tag-derived constants exist.  Domains currently built on the
HIP path: cartpole, cheetah, humanoid plus walker, pendulum,
acrobot, hopper, reacher and point_mass (first SURVEY.md 8f row); the remaining reference
domains need primitives that are not implemented yet.

Batched use: `environment_kwargs={'batch_size': 8192, 'device': 0,
'precision': 'f32', 'device_init': True}`.
"""

import collections
import inspect

from dm_control_amd.rl import control
from dm_control_amd.suite import acrobot
from dm_control_amd.suite import cartpole
from dm_control_amd.suite import cheetah
from dm_control_amd.suite import hopper
from dm_control_amd.suite import humanoid
from dm_control_amd.suite import pendulum
from dm_control_amd.suite import point_mass
from dm_control_amd.suite import reacher
from dm_control_amd.suite import walker

_DOMAINS = {name: module for name, module in locals().items()
            if inspect.ismodule(module) and hasattr(module, 'SUITE')}


def _get_tasks(tag):
  result = []
  for domain_name in sorted(_DOMAINS.keys()):
    domain = _DOMAINS[domain_name]
    tasks = domain.SUITE if tag is None else domain.SUITE.tagged(tag)
    for task_name in tasks.keys():
      result.append((domain_name, task_name))
  return tuple(result)


def _get_tasks_by_domain(tasks):
  result = collections.defaultdict(list)
  for domain_name, task_name in tasks:
    result[domain_name].append(task_name)
  return {k: tuple(v) for k, v in result.items()}


ALL_TASKS = _get_tasks(tag=None)
BENCHMARKING = _get_tasks('benchmarking')
EASY = _get_tasks('easy')
HARD = _get_tasks('hard')
EXTRA = tuple(sorted(set(ALL_TASKS) - set(BENCHMARKING)))
NO_REWARD_VIZ = _get_tasks('no_reward_visualization')
REWARD_VIZ = tuple(sorted(set(ALL_TASKS) - set(NO_REWARD_VIZ)))
TASKS_BY_DOMAIN = _get_tasks_by_domain(ALL_TASKS)


def load(domain_name, task_name, task_kwargs=None, environment_kwargs=None,
         visualize_reward=False):
  """Returns an environment from a domain name, task name and settings."""
  return build_environment(domain_name, task_name, task_kwargs,
                           environment_kwargs, visualize_reward)


def build_environment(domain_name, task_name, task_kwargs=None,
                      environment_kwargs=None, visualize_reward=False):
  if domain_name not in _DOMAINS:
    raise ValueError('Domain {!r} does not exist.'.format(domain_name))
  domain = _DOMAINS[domain_name]
  if task_name not in domain.SUITE:
    raise ValueError('Level {!r} does not exist in domain {!r}.'.format(
        task_name, domain_name))
  task_kwargs = task_kwargs or {}
  if environment_kwargs is not None:
    task_kwargs = dict(task_kwargs, environment_kwargs=environment_kwargs)
  env = domain.SUITE[task_name](**task_kwargs)
  env.task.visualize_reward = visualize_reward
  return env
