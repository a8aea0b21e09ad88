"""Control Suite loader for the batched MI355X physics step.

Contract of the reference's loader
(/root/reference/dm_control/suite/__init__.py:78-150, loader_test.py:23-42):
`load(domain_name, task_name, task_kwargs, environment_kwargs,
visualize_reward)` returns a `control.Environment`; unknown names raise
ValueError; the tag-derived task tuples exist under the same names.

Domains on the HIP path: cartpole, cheetah, humanoid and walker,
pendulum, acrobot, hopper, reacher, point_mass.  The other
reference domains need physics that is not built.

Batched use: `environment_kwargs={'batch_size': 8192, 'device': 0,
'precision': 'f32', 'device_init': True}`.
"""

import importlib

from dm_control_amd.rl import control

# The registry is an explicit table: one row per domain module on the HIP path.
DOMAIN_NAMES = ('acrobot', 'cartpole', 'cheetah', 'hopper', 'humanoid',
                'pendulum', 'point_mass', 'reacher', 'walker')


class _Registry:
  """domain name -> module, with the task tables derived once at import."""

  def __init__(self, names):
    self.modules = {}
    for name in names:
      module = importlib.import_module('dm_control_amd.suite.' + name)
      if not hasattr(module, 'SUITE'):
        raise ImportError('suite domain %r defines no SUITE task table' % name)
      self.modules[name] = module
      globals()[name] = module        # `suite.cheetah` etc. stay importable

  def pairs(self, tag=None):
    """(domain, task) pairs in domain-then-registration order."""
    out = []
    for name in sorted(self.modules):
      table = self.modules[name].SUITE
      chosen = table if tag is None else table.tagged(tag)
      out.extend((name, task) for task in chosen)
    return tuple(out)

  def factory(self, domain_name, task_name):
    module = self.modules.get(domain_name)
    if module is None:
      raise ValueError('Domain {!r} does not exist.'.format(domain_name))
    if task_name not in module.SUITE:
      raise ValueError('Level {!r} does not exist in domain {!r}.'.format(
          task_name, domain_name))
    return module.SUITE[task_name]


_REGISTRY = _Registry(DOMAIN_NAMES)
_DOMAINS = _REGISTRY.modules


def _without(pairs, excluded):
  drop = set(excluded)
  return tuple(sorted(p for p in pairs if p not in drop))


ALL_TASKS = _REGISTRY.pairs()
BENCHMARKING = _REGISTRY.pairs('benchmarking')
EASY = _REGISTRY.pairs('easy')
HARD = _REGISTRY.pairs('hard')
NO_REWARD_VIZ = _REGISTRY.pairs('no_reward_visualization')
EXTRA = _without(ALL_TASKS, BENCHMARKING)
REWARD_VIZ = _without(ALL_TASKS, NO_REWARD_VIZ)
TASKS_BY_DOMAIN = {
    domain: tuple(task for d, task in ALL_TASKS if d == domain)
    for domain in sorted(_DOMAINS)}


def build_environment(domain_name, task_name, task_kwargs=None,
                      environment_kwargs=None, visualize_reward=False):
  """Looks the task factory up and calls it (suite/__init__.py:117-150)."""
  factory = _REGISTRY.factory(domain_name, task_name)
  kwargs = dict(task_kwargs or {})
  if environment_kwargs is not None:
    kwargs['environment_kwargs'] = environment_kwargs
  env = factory(**kwargs)
  if not isinstance(env, control.Environment):
    raise TypeError('task factory {}.{} returned {!r}'.format(
        domain_name, task_name, type(env)))
  env.task.visualize_reward = visualize_reward
  return env


def load(domain_name, task_name, task_kwargs=None, environment_kwargs=None,
         visualize_reward=False):
  """`suite.load` of the reference (suite/__init__.py:93-114)."""
  return build_environment(domain_name, task_name, task_kwargs=task_kwargs,
                           environment_kwargs=environment_kwargs,
                           visualize_reward=visualize_reward)
