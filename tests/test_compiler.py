"""MJCF-subset compiler: derived checks that do not need libmujoco."""

import math
import os

import numpy as np
import pytest

import helpers
import kat_models
from dm_control_amd import codegen
from dm_control_amd.mjcf import compiler
from dm_control_amd.mjcf import model as mdl
from oracle import oracle

REF_SUITE = '/root/reference/dm_control/suite'


@pytest.mark.parametrize('name', ['cartpole', 'cheetah', 'humanoid', 'walker',
                                  'pendulum', 'acrobot', 'hopper', 'reacher',
                                  'point_mass'])
def test_in_tree_models_compile_like_the_reference_files(name):
  """The in-tree parameter tables (suite/models/*.py) give the same compiled
  model, names and ordering included, as the reference's MJCF files (which also
  carry rendering includes, sites and sensors no task on the path reads)."""
  path = os.path.join(REF_SUITE, name + '.xml')
  if not os.path.exists(path):
    pytest.skip('reference tree not present')
  a = compiler.from_xml_path(path)
  b = helpers.load_model(name)
  skip = {'nsensor', 'nsensordata', 'sensor_type', 'sensor_objid',
          'sensor_adr', 'sensor_dim'}
  for field, _ in mdl.FIELDS:
    if field in skip:
      continue
    if name in ('cartpole', 'pendulum') and (field.startswith('geom_') or field in (
        'ngeom', 'body_geomnum', 'body_geomadr')):
      continue   # decorative floor/rails dropped; contacts are disabled
    np.testing.assert_array_equal(np.asarray(a.field(field)),
                                  np.asarray(b.field(field)), err_msg=field)
  for kind in ('body', 'joint', 'actuator'):
    assert a.names[kind] == b.names[kind]


def test_sizes_of_the_three_target_models():
  """SURVEY.md 8(a) sizes."""
  cp, ch, hu = (helpers.load_model(n) for n in ('cartpole', 'cheetah',
                                                'humanoid'))
  assert (cp.nbody, cp.nq, cp.nv, cp.nu) == (3, 2, 2, 1)
  assert (ch.nbody, ch.nq, ch.nv, ch.nu, ch.ngeom) == (8, 9, 9, 6, 9)
  assert (hu.nbody, hu.nq, hu.nv, hu.nu, hu.ngeom) == (17, 28, 27, 21, 20)
  assert codegen.observation_size(cp, codegen.TASK_CARTPOLE) == 5
  assert codegen.observation_size(ch, codegen.TASK_CHEETAH) == 17
  assert codegen.observation_size(hu, codegen.TASK_HUMANOID) == 67
  np.testing.assert_allclose(ch.body_mass.sum(), 14.0)   # settotalmass
  assert cp.opt.integrator == mdl.INT_RK4 and ch.opt.integrator == mdl.INT_EULER
  assert cp.opt.disableflags & mdl.DSBL_CONTACT


def test_capsule_inertia_closed_form():
  """SURVEY.md Appendix D."""
  r, h, rho = 0.045, 0.5, 1000.0
  m = compiler.from_xml_string(
      '<mujoco><worldbody><body><joint type="hinge"/>'
      '<geom type="capsule" size="%g %g"/></body></worldbody></mujoco>' % (r, h))
  m_c, m_s = rho*math.pi*r*r*2*h, rho*4/3*math.pi*r**3
  izz = m_c*r*r/2 + 2*m_s*r*r/5
  ixx = m_c*(3*r*r + (2*h)**2)/12 + m_s*(2*r*r/5 + h*h + 3*h*r/4)
  np.testing.assert_allclose(m.body_mass[1], m_c + m_s, rtol=1e-14)
  np.testing.assert_allclose(m.body_inertia[1], [ixx, ixx, izz], rtol=1e-13)


def test_defaults_classes_and_freejoint():
  hu = helpers.load_model('humanoid')
  root = hu.name2id('root', 'joint')
  assert hu.jnt_type[root] == mdl.JNT_FREE
  # <freejoint> takes no defaults; class "body" joints do
  assert hu.dof_damping[:6].tolist() == [0]*6 and not hu.jnt_limited[root]
  j = hu.name2id('abdomen_z', 'joint')
  assert hu.jnt_stiffness[j] == 20 and hu.dof_damping[hu.jnt_dofadr[j]] == 5
  np.testing.assert_allclose(hu.jnt_range[j], np.deg2rad([-45, 45]))
  np.testing.assert_allclose(hu.jnt_solimp[j], [0, .99, .01, .5, 2])
  ch = helpers.load_model('cheetah')
  np.testing.assert_allclose(ch.jnt_range[ch.name2id('fthigh', 'joint')],
                             np.deg2rad([-57, .40]))
  assert ch.actuator_ctrllimited.all()
  with pytest.raises(ValueError):
    hu.name2id('nope', 'joint')


def test_fixed_tendons_and_tendon_actuators():
  """point_mass.xml: two fixed tendons over the sliders, motors on tendons."""
  pm = helpers.load_model('point_mass')
  assert (pm.ntendon, pm.nwrap) == (2, 4)
  assert pm.tendon_adr.tolist() == [0, 2] and pm.tendon_num.tolist() == [2, 2]
  assert pm.wrap_objid.tolist() == [0, 1, 0, 1]
  assert pm.wrap_prm.tolist() == [1, 0, 0, 1]
  assert pm.actuator_trntype.tolist() == [mdl.TRN_TENDON]*2
  assert pm.actuator_trnid.tolist() == [0, 1]
  # oracle: control 0 accelerates x only; with swapped coefficients y only
  d = oracle.OracleData(oracle.OracleModel(pm))
  d.ctrl[:] = [1, 0]
  d.step1()
  d.physics_step()
  assert d.qvel[0] > 0 and d.qvel[1] == 0
  import copy
  sw = copy.copy(pm)
  sw.wrap_prm = np.array([0.0, 1, 1, 0])
  d = oracle.OracleData(oracle.OracleModel(sw))
  d.ctrl[:] = [1, 0]
  d.step1()
  d.physics_step()
  assert d.qvel[1] > 0 and d.qvel[0] == 0


def test_unsupported_features_raise():
  with pytest.raises(compiler.CompileError):
    compiler.from_xml_string(      # spatial tendons are not implemented
        '<mujoco><worldbody/><tendon><spatial/></tendon></mujoco>')
  with pytest.raises(compiler.CompileError):
    compiler.from_xml_string(      # nor are tendon springs / dampers / limits
        '<mujoco><worldbody><body><joint name="j"/><geom size="1"/></body>'
        '</worldbody><tendon><fixed stiffness="2"><joint joint="j" coef="1"/>'
        '</fixed></tendon></mujoco>')
  with pytest.raises(compiler.CompileError):
    compiler.from_xml_string(
        '<mujoco><worldbody><body><joint frictionloss="1"/><geom size="1"/>'
        '</body></worldbody></mujoco>')
  with pytest.raises(compiler.CompileError):
    compiler.from_xml_string('<mujoco><include file="missing.xml"/></mujoco>')
  with pytest.raises(compiler.CompileError):
    compiler.from_xml_string('<notmujoco/>')
  m = compiler.from_xml_string(
      '<mujoco><worldbody><geom type="plane" size="1 1 1"/><body pos="0 0 1">'
      '<freejoint/><geom type="cylinder" size=".1 .1"/></body></worldbody>'
      '</mujoco>')
  with pytest.raises(codegen.UnsupportedModelError):
    codegen.collision_pairs(m)


def test_include_and_assets():
  inc = '<mujoco><worldbody><body name="b"><joint type="slide"/>' \
        '<geom size=".1"/></body></worldbody></mujoco>'
  m = compiler.from_xml_string('<mujoco><include file="./x/inc.xml"/></mujoco>',
                               assets={'./x/inc.xml': inc.encode()})
  assert m.nbody == 2 and m.nv == 1


@pytest.mark.parametrize('name', ['cartpole', 'cheetah', 'humanoid'])
def test_mass_matrix_two_ways(name):
  """Compiler's body-Jacobian mass matrix == oracle's CRBA mass matrix."""
  m = helpers.load_model(name)
  mm, _ = compiler.mass_matrix_qpos0(m)
  p = oracle.OraclePhysics(m)
  p.reset()
  np.testing.assert_allclose(p.data.qM, mm, rtol=1e-12, atol=1e-12)
  np.testing.assert_allclose(m.meaninertia, np.mean(np.diag(mm)))


def test_cartpole_mass_matrix_textbook():
  """cart 1 kg, pole 0.1 kg with CoM at 0.5 m."""
  m = helpers.load_model('cartpole')
  p = oracle.OraclePhysics(m)
  p.reset()
  theta = 0.7
  p.data.qpos[:] = [0.2, theta]
  p.forward()
  mc, mp, l = 1.0, 0.1, 0.5
  ipole = m.body_inertia[2][0]
  expect = np.array([[mc + mp, mp*l*math.cos(theta)],
                     [mp*l*math.cos(theta), ipole + mp*l*l]])
  np.testing.assert_allclose(p.data.qM, expect, rtol=1e-12)
  # gravity torque on the hinge: qfrc_bias[1] = -m g l sin(theta)
  np.testing.assert_allclose(p.data.qfrc_bias[1],
                             -mp*9.81*l*math.sin(theta), rtol=1e-12)


def test_static_pairs_and_mixing():
  ch = helpers.load_model('cheetah')
  pairs = codegen.collision_pairs(ch)
  assert len(pairs) == 8 + 19      # floor x 8 capsules + non-adjacent pairs
  mx = codegen.mix_pair(ch, *pairs[0])
  assert mx['dim'] == 3 and mx['friction'][0] == 1.0   # max(.4, 1)
  hu = helpers.load_model('humanoid')
  floor = [p for p in codegen.collision_pairs(hu)
           if hu.geom_type[p[0]] == mdl.GEOM_PLANE]
  mx = codegen.mix_pair(hu, *floor[0])
  assert mx['dim'] == 3                                 # max(condim 3, 1)
  np.testing.assert_allclose(mx['solref'], [0.0175, 1.0])
  np.testing.assert_allclose(mx['solimp'][:3], [0.9, 0.97, 0.002])
  assert helpers.load_model('cartpole').opt.disableflags & mdl.DSBL_CONTACT
  assert codegen.collision_pairs(helpers.load_model('cartpole')) == []


def test_generated_header_mentions_every_table():
  text = codegen.generate_header(helpers.load_model('cheetah'),
                                 codegen.TASK_CHEETAH)
  for token in ('NEFC_MAX', 'pair_g1', 'limit_K', 'body_parentid',
                'DMC_UNROLL', 'task_body'):
    assert token in text
  m = compiler.from_xml_string(kat_models.PRIMITIVES)
  assert codegen.model_info(m)['npair'] > 0
