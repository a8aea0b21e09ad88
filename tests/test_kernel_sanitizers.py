"""Sanitizer runs of the kernel SOURCES on the host (no GPU needed).

GPU AddressSanitizer is not available, so csrc/dmc_kernels.hip is compiled as
plain C++ through tests/host_shim/shim.h (one lane, one workgroup, fp64) with
-fsanitize=address,undefined and stepped next to the oracle.  This checks the
kernel's indexing (static chains, LDS/HBM record tiers, contact list) for
out-of-bounds and UB, and its arithmetic against the oracle, before anything
touches the card.  The shim is test infrastructure: the product path cannot
reach it.
"""

import os
import subprocess

import numpy as np
import pytest

import helpers
import kat_models
from dm_control_amd import codegen
from dm_control_amd.mjcf import compiler
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, 'tests', 'host_shim')
KERNEL = os.path.join(ROOT, 'dm_control_amd', 'csrc', 'dmc_kernels.hip')


def _build(model, task, tmp_path, unroll, extra=(), sanitize=True, f64=True,
           name='harness'):
  header = tmp_path/'model.h'
  text = codegen.generate_header(model, task, unroll=unroll)
  header.write_text(text.replace('static __device__ constexpr',
                                 'static constexpr'))
  exe = tmp_path/name
  mode = (['-O1', '-g', '-fsanitize=address,undefined',
           '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer']
          if sanitize else ['-O2', '-ffp-contract=off'])
  cmd = ['g++', '-std=c++17', '-w'] + mode + (
      ['-DDMC_REAL_IS_DOUBLE'] if f64 else []) + [
         '-DDMC_LDS_BUDGET=16384'] + list(extra) + [
         '-DDMC_MODEL_HEADER="%s"' % header,
         '-DDMC_KERNEL_SOURCE="%s"' % KERNEL,
         '-I', os.path.join(ROOT, 'dm_control_amd', 'csrc'), '-I', SHIM,
         '-x', 'c++', os.path.join(SHIM, 'harness.cpp'), '-o', str(exe)]
  subprocess.check_call(cmd)
  return str(exe)


def _run(exe, steps, qpos, qvel):
  args = [exe, str(steps)] + ['%.17g' % v for v in qpos] + \
         ['%.17g' % v for v in qvel]
  env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0')
  out = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, env=env, timeout=600)
  assert out.returncode == 0, out.stderr[-3000:]
  rows = []
  for line in out.stdout.splitlines():
    if line.startswith('STEP'):
      vals, tail = line.split('|')
      rows.append((np.array([float(v) for v in vals.split()[2:]]),
                   [int(v) for v in tail.split()]))
  return rows


@pytest.mark.timeout(900)
@pytest.mark.parametrize('name,unroll,extra', [
    ('cheetah', True, ()), ('cheetah', False, ()), ('primitives', True, ()),
    ('hopper', True, ()),                            # touch sensors
    # precision='mixed' (DMC_STATE_COMP): the (high, low) state words and their
    # workspace tier under ASan; with real = double the low words are zero and
    # the trajectory must still be the oracle's
    ('cheetah', True, ('-DDMC_STATE_COMP=1',)),
    ('cartpole', True, ('-DDMC_STATE_COMP=1',)),
    # the packed matrices in the HBM workspace (the path of scenes with several
    # walkers, forced here on small models)
    ('cheetah', False, ('-DDMC_MAT_PRIVATE_BYTES=64',)),
    ('primitives', False, ('-DDMC_MAT_PRIVATE_BYTES=64',)),
    # box-box (face contacts), capsule-box and plane-box in one stack
    ('stacked_boxes', True, ()), ('stacked_boxes', False, ())])
def test_kernel_source_is_clean_and_matches_oracle(name, unroll, extra, tmp_path):
  if name == 'primitives':
    model, task = compiler.from_xml_string(kat_models.PRIMITIVES), 0
    qpos, qvel = model.qpos0.copy(), np.zeros(model.nv)
    qpos[2], qpos[9], qpos[16] = 0.11, 0.2, 0.3     # stacked, in contact
    steps = 40
  elif name == 'stacked_boxes':
    model, task = compiler.from_xml_string(kat_models.STACKED_BOXES), 0
    qpos, qvel = model.qpos0.copy(), np.zeros(model.nv)
    qpos[7 + 3:7 + 7] = [0.98, 0.05, -0.1, 0.15]     # tilt one box: edge contacts too
    qpos[7 + 3:7 + 7] /= np.linalg.norm(qpos[7 + 3:7 + 7])
    qvel[:] = 0.3*np.random.RandomState(5).randn(model.nv)
    steps = 60
  else:
    model, task = helpers.load_model(name), helpers.TASKS[name]
    q, v = helpers.initial_states(model, name, 4, seed=7)
    qpos, qvel = q[1], v[1]
    steps = 25
  exe = _build(model, task, tmp_path, unroll, extra)
  rows = _run(exe, steps, qpos, qvel)
  assert len(rows) == steps
  d = oracle.OracleData(oracle.OracleModel(model))
  d.qpos[:] = qpos
  d.qvel[:] = qvel
  d.step1()
  touched = False
  for state, (ncon, nefc, iters, warn) in rows:
    touched |= d.nefc > 0
    # contact and row counts as mj_makeConstraint counts them (a pyramid edge
    # pair that the planar models store as one row still counts as two)
    assert (ncon, nefc) == (d.ncon, d.nefc)
    d.physics_step()
    assert warn == 0
    np.testing.assert_allclose(state[:model.nq], d.qpos, rtol=0, atol=1e-9)
    np.testing.assert_allclose(state[model.nq:], d.qvel, rtol=0, atol=1e-8)
  # constraint rows (LDS and HBM tiers) were exercised (the cart-pole: RK4 path)
  assert touched or name == 'cartpole'


@pytest.mark.timeout(900)
def test_mixed_precision_source_beats_plain_fp32_on_the_smooth_system(tmp_path):
  """precision='mixed' = fp32 arithmetic with qpos/qvel carried between steps
  as fp64 (high, low) pairs (DMC_STATE_COMP).  The kernel source is built for
  the host in fp32 with and without it and run free for 1000 steps of the
  cart-pole next to the fp64 oracle: the compensated state must stay closer
  (state rounding is what the tail of the fp32 build is made of)
  and within BASELINE's 1e-4."""
  model, task = helpers.load_model('cartpole'), helpers.TASKS['cartpole']
  q, v = helpers.initial_states(model, 'cartpole', 8, seed=3)
  exes = {tag: _build(model, task, tmp_path, True, extra, sanitize=False,
                      f64=False, name=tag)
          for tag, extra in (('f32', ()), ('mixed', ('-DDMC_STATE_COMP=1',)))}
  om = oracle.OracleModel(model)
  worse = 0
  for e in range(4):
    d = oracle.OracleData(om)
    d.qpos[:] = q[e]
    d.qvel[:] = v[e]
    d.step1()
    ref = []
    for _ in range(1000):
      d.physics_step()
      ref.append(d.qpos.copy())
    ref = np.array(ref)
    err = {}
    for tag, exe in exes.items():
      st = np.array([r[0][:model.nq] for r in _run(exe, 1000, q[e], v[e])])
      err[tag] = np.abs(st - ref).max(axis=1)/np.maximum(1, np.abs(ref).max(axis=1))
    assert err['mixed'][-1] <= 1e-4
    assert err['mixed'][:100].max() <= 1e-6
    worse += err['mixed'][-1] > err['f32'][-1]
  assert worse <= 1       # rounding is not monotone env by env; the rule is


# ---------------------------------------------------------------------------
# several lanes per env (csrc/dmc_coop.hip): one OS thread per lane
# ---------------------------------------------------------------------------
COOP_KERNEL = os.path.join(ROOT, 'dm_control_amd', 'csrc', 'dmc_coop.hip')


def _build_coop(model, task, tmp_path, sanitizer, group):
  header = tmp_path/'model.h'
  text = codegen.generate_header(model, task, unroll=True)
  header.write_text(text.replace('static __device__ constexpr',
                                 'static constexpr'))
  exe = tmp_path/'harness_coop'
  cmd = ['g++', '-std=c++17', '-O1', '-g', '-pthread',
         '-fsanitize=' + sanitizer, '-fno-omit-frame-pointer',
         '-DDMC_REAL_IS_DOUBLE', '-DDMC_GROUP=%d' % min(group, 64),
         '-DDMC_COOP_DUO=%d' % (group == 128),
         '-DDMC_MODEL_HEADER="%s"' % header,
         '-DDMC_KERNEL_SOURCE="%s"' % COOP_KERNEL,
         '-I', os.path.join(ROOT, 'dm_control_amd', 'csrc'), '-I', SHIM,
         '-x', 'c++', os.path.join(SHIM, 'harness_coop.cpp'), '-o', str(exe)]
  if 'undefined' in sanitizer:
    cmd.insert(1, '-fno-sanitize-recover=undefined')
  subprocess.check_call(cmd)
  return str(exe)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize('name,sanitizer,group,steps', [
    ('humanoid', 'address,undefined', 64, 9),
    ('cheetah', 'thread', 32, 12),
    ('primitives', 'thread', 32, 20),
    ('cartpole', 'address,undefined', 64, 6),
    ('hopper', 'address,undefined', 64, 12),
    # two wavefronts per env: the row-building wave runs concurrently with the
    # mass-matrix / velocity wave, ThreadSanitizer watches their LDS regions
    ('humanoid', 'thread', 128, 9),
    ('hopper', 'thread', 128, 10),
    ('humanoid', 'address,undefined', 128, 9),
    ('stacked_boxes', 'address,undefined', 64, 30)])
def test_several_lanes_per_env_source(name, sanitizer, group, steps, tmp_path):
  """csrc/dmc_coop.hip with one thread per lane (tests/host_shim/shim_coop.h):
  a phase hand-over is a pthread barrier, so ThreadSanitizer reports any LDS
  word that crosses lanes without one, AddressSanitizer every index; the
  trajectories of all envs of the workgroup are compared with the oracle."""
  nenv = max(1, 64//group)      # group 128: one env, two wavefronts
  if name == 'primitives':
    model, task = compiler.from_xml_string(kat_models.PRIMITIVES), 0
    q = np.tile(model.qpos0, (nenv, 1))
    v = np.zeros((nenv, model.nv))
    q[:, 2], q[:, 9], q[:, 16] = 0.11, 0.2, 0.3
  elif name == 'stacked_boxes':
    model, task = compiler.from_xml_string(kat_models.STACKED_BOXES), 0
    q = np.tile(model.qpos0, (nenv, 1))
    v = 0.3*np.random.RandomState(5).randn(nenv, model.nv)
  else:
    model, task = helpers.load_model(name), helpers.TASKS[name]
    q, v = helpers.initial_states(model, name, max(nenv, 2), seed=7)
    q, v = q[-nenv:], v[-nenv:]
  exe = _build_coop(model, task, tmp_path, sanitizer, group)
  args = [exe, str(steps), '1']
  for e in range(nenv):
    args += ['%.17g' % x for x in q[e]] + ['%.17g' % x for x in v[e]]
  env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0',
             TSAN_OPTIONS='halt_on_error=1')
  out = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, env=env, timeout=1100)
  assert out.returncode == 0, out.stderr[-3000:]
  om = oracle.OracleModel(model)
  datas = []
  for e in range(nenv):
    d = oracle.OracleData(om)
    d.qpos[:] = q[e]
    d.qvel[:] = v[e]
    d.step1()
    datas.append(d)
  seen, touched = 0, False
  for line in out.stdout.splitlines():
    if not line.startswith('STEP'):
      continue
    vals, tail = line.split('|')
    fields = vals.split()
    e = int(fields[2])
    state = np.array([float(x) for x in fields[3:]])
    ncon, nefc, iters, warn = [int(x) for x in tail.split()]
    d = datas[e]
    touched |= d.nefc > 0
    assert (ncon, nefc) == (d.ncon, d.nefc)
    d.physics_step()
    assert warn == 0
    np.testing.assert_allclose(state[:model.nq], d.qpos, rtol=0, atol=1e-9)
    np.testing.assert_allclose(state[model.nq:], d.qvel, rtol=0, atol=1e-8)
    seen += 1
  assert seen == steps*nenv
  assert touched or name == 'cartpole'


# ---------------------------------------------------------------------------
# team mode of csrc/dmc_kernels.hip (one wavefront per env, big scenes): one OS
# thread per lane, a phase boundary (tsync) is a pthread barrier
# ---------------------------------------------------------------------------
def _build_team(model, tmp_path, sanitizer, team, ncon_max=64):
  header = tmp_path/'model.h'
  text = codegen.generate_header(model, 0, ncon_max=ncon_max, unroll=False)
  header.write_text(text.replace('static __device__ constexpr',
                                 'static constexpr'))
  exe = tmp_path/'harness_team'
  cmd = ['g++', '-std=c++17', '-w', '-O1', '-g', '-pthread',
         '-fsanitize=' + sanitizer, '-fno-omit-frame-pointer',
         '-DDMC_REAL_IS_DOUBLE', '-DDMC_TEAM=%d' % team,
         '-DDMC_MODEL_HEADER="%s"' % header,
         '-DDMC_KERNEL_SOURCE="%s"' % KERNEL,
         '-I', os.path.join(ROOT, 'dm_control_amd', 'csrc'), '-I', SHIM,
         '-x', 'c++', os.path.join(SHIM, 'harness_team.cpp'), '-o', str(exe)]
  if 'undefined' in sanitizer:
    cmd.insert(1, '-fno-sanitize-recover=undefined')
  subprocess.check_call(cmd)
  return str(exe)


def _team_scene(name):
  """Scenes of several humanoids and a ball with contacts BETWEEN kinematic
  trees (the coupled blocks of the Hessian): two walkers pushed into each other
  with the ball between their feet; four walkers in a tangle."""
  from dm_control_amd.locomotion.models import soccer
  rs = np.random.RandomState(3)
  if name == 'two_walkers_touching':
    m = compiler.from_xml_string(soccer.build(2, with_ball=True, ball=soccer.REGULATION_BALL))
  else:
    m = compiler.from_xml_string(soccer.build(4, with_ball=True, ball=soccer.REGULATION_BALL,
                                              goal_size=(1.0, 1.6, 1.0)))
  nw = (m.nq - 7)//63
  qpos = np.array(m.qpos0, float)
  for k in range(nw):
    qpos[63*k + 2] = 0.86 + 0.02*k          # feet in the ground
    qpos[63*k + 7:63*k + 63] += 0.3*rs.randn(56)
  if name == 'two_walkers_touching':
    qpos[63:65] = qpos[0:2] + [0.25, 0.0]
    qpos[126:129] = [qpos[0] + 0.1, qpos[1], 0.10]
  else:
    for k, (x, y) in enumerate([(0, 0), (0.3, 0), (0, 0.35), (0.3, 0.35)]):
      qpos[63*k:63*k + 2] = [x, y]
    qpos[252:255] = [0.15, 0.17, 0.45]
  return m, qpos, 0.5*rs.randn(m.nv)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('name,sanitizer,team,steps', [
    ('two_walkers_touching', 'address,undefined', 8, 3),
    ('two_walkers_touching', 'thread', 8, 2),
    ('four_walkers_tangled', 'address,undefined', 8, 2),
    ('four_walkers_tangled', 'thread', 4, 1)])
def test_team_build_of_a_scene_with_several_trees(name, sanitizer, team, steps, tmp_path):
  """csrc/dmc_kernels.hip in team mode (-DDMC_TEAM: the lanes of a wavefront
  share ONE env -- what `locomotion.soccer` runs) with one thread per lane:
  shared LDS / workspace words have one writer per phase (ThreadSanitizer),
  every index is in range (AddressSanitizer), and the trajectory is the
  oracle's -- with contacts between trees, so the coupled blocks of the Newton
  Hessian (rows left of a tree's tile) are factored and solved too."""
  m, qpos, qvel = _team_scene(name)
  exe = _build_team(m, tmp_path, sanitizer, team)
  args = [exe, str(steps)] + ['%.17g' % x for x in qpos] + ['%.17g' % x for x in qvel]
  env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0', TSAN_OPTIONS='halt_on_error=1')
  out = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, env=env, timeout=1400)
  assert out.returncode == 0, out.stderr[-3000:]
  om = oracle.OracleModel(m)
  d = oracle.OracleData(om)
  d.qpos[:] = qpos
  d.qvel[:] = qvel
  d.step1()
  seen, coupled = 0, False
  tree_of_dof = np.asarray(m.body_rootid)[np.asarray(m.dof_bodyid)]
  for line in out.stdout.splitlines():
    if not line.startswith('STEP'):
      continue
    vals, tail = line.split('|')
    state = np.array([float(x) for x in vals.split()[2:]])
    ncon, nefc, iters, warn = [int(x) for x in tail.split()]
    assert (ncon, nefc) == (d.ncon, d.nefc)
    J = np.asarray(d.efc_J_matrix())[:d.nefc] if callable(getattr(d, 'efc_J_matrix', None)) else None
    if J is not None:
      coupled |= any(len(set(tree_of_dof[np.nonzero(row)[0]])) > 1 for row in J)
    d.physics_step()
    assert warn == 0
    np.testing.assert_allclose(state[:m.nq], d.qpos, rtol=0, atol=1e-9)
    np.testing.assert_allclose(state[m.nq:], d.qvel, rtol=0, atol=1e-8)
    seen += 1
  assert seen == steps
  assert coupled, 'no constraint row touched two trees: the scene does not test the coupling'
