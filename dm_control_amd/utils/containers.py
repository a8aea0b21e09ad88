"""`TaggedTasks` registry (cf. /root/reference/dm_control/utils/containers.py)."""

import collections


class TaggedTasks(collections.abc.Mapping):
  """Maps task names to factory functions; factories can carry tags."""

  def __init__(self, allow_overriding_keys=False):
    self._tasks = collections.OrderedDict()
    self._tags = collections.defaultdict(list)
    self._allow_overriding_keys = allow_overriding_keys

  def add(self, *tags):
    def wrap(factory):
      name = factory.__name__
      if name in self._tasks and not self._allow_overriding_keys:
        raise ValueError('Function named {!r} already exists in the container '
                         'and `allow_overriding_keys` is False.'.format(name))
      self._tasks[name] = factory
      for tag in tags:
        self._tags[tag].append(name)
      return factory
    return wrap

  def tagged(self, *tags):
    if not tags:
      return {}
    names = None
    for tag in tags:
      if tag not in self._tags:
        return {}
      these = set(self._tags[tag])
      names = these if names is None else names & these
    return collections.OrderedDict(
        (k, v) for k, v in self._tasks.items() if k in names)

  def tags(self):
    return list(self._tags.keys())

  def __getitem__(self, k):
    return self._tasks[k]

  def __iter__(self):
    return iter(self._tasks)

  def __len__(self):
    return len(self._tasks)
