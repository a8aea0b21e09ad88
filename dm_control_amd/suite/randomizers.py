"""Initial-state randomisers (cf. suite/utils/randomizers.py:35-86)."""

import numpy as np

from dm_control_amd.mjcf import model as mdl


def randomize_limited_and_rotational_joints(model, qpos, random):
  """The reference's rules applied to a plain qpos vector, same draw order.

  Bounded hinges/sliders: uniform in range; unbounded hinges: uniform in
  [-pi, pi]; free joints: only the quaternion (normalised `rand(4)`); ball
  joints: normalised `randn(4)`.
  """
  for j in range(model.njnt):
    jtype = model.jnt_type[j]
    a = model.jnt_qposadr[j]
    lo, hi = model.jnt_range[j]
    if model.jnt_limited[j]:
      if jtype in (mdl.JNT_HINGE, mdl.JNT_SLIDE):
        qpos[a] = random.uniform(lo, hi)
    else:
      if jtype == mdl.JNT_HINGE:
        qpos[a] = random.uniform(-np.pi, np.pi)
      elif jtype == mdl.JNT_BALL:
        quat = random.randn(4)
        qpos[a:a + 4] = quat/np.linalg.norm(quat)
      elif jtype == mdl.JNT_FREE:
        quat = random.rand(4)
        qpos[a + 3:a + 7] = quat/np.linalg.norm(quat)


def randomized_qpos(task, physics):
  """[B, nq] initial positions, one RandomState stream per instance."""
  rows = []
  for rs in task.streams(physics):
    qpos = physics.model.qpos0.copy()
    randomize_limited_and_rotational_joints(physics.model, qpos, rs)
    rows.append(qpos)
  return np.array(rows)
