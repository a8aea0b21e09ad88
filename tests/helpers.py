"""Shared helpers for the parity tests (oracle = checker, HIP path = product)."""

import os

import numpy as np

from dm_control_amd import codegen
from dm_control_amd.mjcf import compiler

TASKS = {'cartpole': codegen.TASK_CARTPOLE, 'cheetah': codegen.TASK_CHEETAH,
         'humanoid': codegen.TASK_HUMANOID, 'walker': codegen.TASK_WALKER,
         'pendulum': codegen.TASK_PENDULUM, 'acrobot': codegen.TASK_ACROBOT,
         'hopper': codegen.TASK_HOPPER, 'reacher': codegen.TASK_REACHER,
         'point_mass': codegen.TASK_POINTMASS}
# build mode per suite model (humanoid: see suite/humanoid.py)
MODES = {'cartpole': 'auto', 'cheetah': 'auto', 'humanoid': 'coop',
         'walker': 'auto', 'pendulum': 'auto', 'acrobot': 'auto',
         'hopper': 'auto', 'reacher': 'auto',
         'point_mass': 'auto'}


def model_xml(name):
  from dm_control_amd.suite import common
  return common.read_model(name + '.xml')


def load_model(name):
  return compiler.from_xml_string(model_xml(name))


def initial_states(model, name, nenv, seed):
  """Plausible, contact-rich initial states [nenv, nq], [nenv, nv]."""
  rs = np.random.RandomState(seed)
  qpos = np.tile(model.qpos0, (nenv, 1))
  qvel = np.zeros((nenv, model.nv))
  if name == 'cartpole':
    qpos[:, 0] = rs.uniform(-1.0, 1.0, nenv)
    qpos[:, 1] = rs.uniform(-np.pi, np.pi, nenv)
    qvel[:] = rs.randn(nenv, model.nv)
  elif name == 'cheetah':
    lim = model.jnt_limited.astype(bool)
    lo, hi = model.jnt_range[lim].T
    qpos[:, lim] = rs.uniform(lo, hi, (nenv, lim.sum()))
    qpos[:, 1] = rs.uniform(-0.1, 0.3, nenv)
    qpos[:, 2] = rs.uniform(-0.5, 0.5, nenv)
    qvel[:] = 0.5*rs.randn(nenv, model.nv)
  elif name == 'pendulum':
    qpos[:, 0] = rs.uniform(-np.pi, np.pi, nenv)
    qvel[:] = rs.randn(nenv, 1)
  elif name == 'hopper':
    lim = model.jnt_limited.astype(bool)
    lo, hi = model.jnt_range[lim].T
    qpos[:, lim] = rs.uniform(0.3*lo, 0.3*hi, (nenv, lim.sum()))
    qpos[:, 1] = rs.uniform(-0.16, -0.04, nenv)   # rootz: foot near / into the floor
    qpos[:, 2] = rs.uniform(-0.2, 0.2, nenv)
    qvel[:] = 0.5*rs.randn(nenv, model.nv)
  elif name == 'point_mass':
    qpos[:] = rs.uniform(-0.29, 0.29, (nenv, 2))   # incl. at the joint limits
    qpos[::5] *= 1.02
    qvel[:] = 0.3*rs.randn(nenv, 2)
  elif name == 'reacher':
    qpos[:, 0] = rs.uniform(-np.pi, np.pi, nenv)
    qpos[:, 1] = rs.uniform(-2.7, 2.7, nenv)
    qvel[:] = 3*rs.randn(nenv, 2)
  elif name == 'acrobot':
    qpos[:] = rs.uniform(-np.pi, np.pi, (nenv, 2))
    qvel[:] = 2*rs.randn(nenv, 2)
  elif name == 'walker':
    lim = model.jnt_limited.astype(bool)
    lo, hi = model.jnt_range[lim].T
    qpos[:, lim] = rs.uniform(0.6*lo, 0.6*hi, (nenv, lim.sum()))
    qpos[:, 0] = rs.uniform(-0.35, 0.05, nenv)    # rootz: feet near/into the floor
    qpos[:, 2] = rs.uniform(-0.4, 0.4, nenv)
    qvel[:] = 0.5*rs.randn(nenv, model.nv)
  elif name == 'humanoid':
    for j in range(model.njnt):
      if model.jnt_limited[j]:
        a = model.jnt_qposadr[j]
        lo, hi = model.jnt_range[j]
        qpos[:, a] = rs.uniform(0.5*lo, 0.5*hi, nenv)
    quat = rs.randn(nenv, 4)*0.3 + np.array([1.0, 0, 0, 0])
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    qpos[:, 3:7] = quat
    qpos[:, 2] = rs.uniform(0.9, 1.5, nenv)
    qvel[:] = 0.3*rs.randn(nenv, model.nv)
  return qpos, qvel


def rel_err(a, b):
  """max_i |a_i - b_i| / max(1, max_i |b_i|), per row."""
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  num = np.max(np.abs(a - b), axis=-1)
  den = np.maximum(1.0, np.max(np.abs(b), axis=-1))
  return num/den


def oracle_touch(model, d, sensor_name):
  """mjSENS_TOUCH from the oracle's contacts and contact forces (call between
  step2 and step1, when the contact list and the forces belong together):
  normal forces of contacts on the site's body whose force ray meets the site
  sphere.  Test-side restatement of mj_sensorAcc's touch case."""
  i = model.names['sensor'].index(sensor_name)
  sid = int(model.sensor_objid[i])
  body = int(model.site_bodyid[sid])
  rot_body = d.xmat[body].reshape(3, 3)
  centre = d.xpos[body] + rot_body.dot(model.site_pos[sid])
  w, x, y, z = model.site_quat[sid]
  rot_site = rot_body.dot(np.array(
      [[w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
       [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
       [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]]))
  size = model.site_size[sid]
  is_box = int(model.site_type[sid]) == 6

  def meets_zone(origin, ray):
    if not is_box:
      b, cc = origin.dot(ray), origin.dot(origin) - size[0]*size[0]
      disc = b*b - cc
      return disc >= 0 and (-b - np.sqrt(disc) >= 0 or -b + np.sqrt(disc) >= 0)
    o, r = rot_site.T.dot(origin), rot_site.T.dot(ray)
    for i in range(3):
      if abs(r[i]) < 1e-15:
        continue
      j, k = (i + 1) % 3, (i + 2) % 3
      for side in (-1, 1):
        t = (side*size[i] - o[i])/r[i]
        if t >= 0 and abs(o[j] + t*r[j]) <= size[j] and abs(o[k] + t*r[k]) <= size[k]:
          return True
    return False
  total = 0.0
  for c in range(d.ncon):
    con = d.contact(c)
    b1 = int(model.geom_bodyid[con['geom1']])
    b2 = int(model.geom_bodyid[con['geom2']])
    if body not in (b1, b2):
      continue
    fn = d.contact_force(c)[0, 0]
    if fn <= 0:
      continue
    ray = con['frame'][0] * (-1.0 if body == b2 else 1.0)
    if meets_zone(con['pos'] - centre, ray):
      total += fn
  return total
