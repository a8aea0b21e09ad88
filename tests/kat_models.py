"""MJCF fixtures of the reference's known-answer tests.

Each string is the model a reference test builds inline; the expected values
live next to the tests that use them.  `GPU_MODELS` are the ones the GPU parity
suite also runs through the HIP kernels (built by `__graft_entry__.build()`).
"""

# K1: /root/reference/dm_control/mujoco/README.md:12-27 (light omitted, colours
# dropped: neither affects the physics)
README_BOX = """
<mujoco>
  <worldbody>
    <geom name="floor" type="plane" size="1 1 .1"/>
    <body name="box" pos="0 0 .3">
      <joint name="up_down" type="slide" axis="0 0 1"/>
      <geom name="box" type="box" size=".2 .2 .2"/>
      <geom name="sphere" pos=".2 .2 .2" size=".1"/>
    </body>
  </worldbody>
</mujoco>
"""

# K2: wrapper/core_test.py:329-344 (touch site/sensor omitted, see the test)
CUBE_ON_FLOOR = """
<mujoco>
  <option gravity="0 0 -9.81"/>
  <worldbody>
    <geom name="floor" type="plane" pos="0 0 0" size="10 10 0.1"/>
    <body name="cube" pos="0 0 0.1">
      <geom type="box" size="0.1 0.1 0.1" mass="1"/>
      <joint type="slide"/>
    </body>
  </worldbody>
</mujoco>
"""

# K3: wrapper/core_test.py:462-472
BOX_ON_FLOOR = """
<mujoco>
  <worldbody>
    <geom name='floor' type='plane' size='1 1 1'/>
    <body name='box' pos='0 0 .1'>
      <freejoint/>
      <geom name='box' type='box' size='.1 .1 .1'/>
    </body>
  </worldbody>
</mujoco>
"""

# K4: wrapper/core_test.py:438-451
CART_POINT_MASS = """
<mujoco>
  <worldbody>
    <body name='cart'>
      <joint type='slide' axis='1 0 0'/>
      <geom name='cart' type='box' size='0.2 0.2 0.2'/>
      <body name='pole'>
        <joint name='hinge' type='hinge' axis='0 1 0'/>
        <geom name='mass' pos='0 0 .5' size='0.04'/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""

# K5: wrapper/core_test.py:509-519
BALL_ON_FLOOR = """
<mujoco>
  <worldbody>
    <geom name='floor' type='plane' size='1 1 1'/>
    <body name='ball' pos='0 0 .1'>
      <freejoint/>
      <geom name='ball' size='.1' friction='1 .1 .1'/>
    </body>
  </worldbody>
</mujoco>
"""

# extra primitives coverage for the device narrowphase: sphere-sphere,
# sphere-capsule and a hinge chain with limits
PRIMITIVES = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="2 2 .1"/>
    <body name="a" pos="0 0 .3">
      <freejoint/>
      <geom name="a" type="sphere" size=".1"/>
    </body>
    <body name="b" pos=".05 0 .6">
      <freejoint/>
      <geom name="b" type="capsule" size=".05 .15" euler="0 70 0"/>
    </body>
    <body name="c" pos="-.05 .02 .9">
      <freejoint/>
      <geom name="c" type="sphere" size=".08"/>
    </body>
    <body name="arm" pos=".6 0 .5">
      <joint name="sh" type="hinge" axis="0 1 0" limited="true" range="-40 40" damping=".05"/>
      <geom name="upper" type="capsule" fromto="0 0 0 .3 0 0" size=".03"/>
      <body name="fore" pos=".3 0 0">
        <joint name="el" type="hinge" axis="0 1 0" limited="true" range="-90 10" damping=".05"/>
        <geom name="lower" type="capsule" fromto="0 0 0 .25 0 0" size=".025"/>
      </body>
    </body>
  </worldbody>
  <actuator>
    <motor name="sh" joint="sh" gear="2" ctrllimited="true" ctrlrange="-1 1"/>
    <motor name="el" joint="el" gear="1" ctrllimited="true" ctrlrange="-1 1"/>
  </actuator>
</mujoco>
"""

# sphere - box: balls dropped onto / rolling against fixed boxes (the soccer
# field box is made of such boxes and only the ball touches it,
# locomotion/soccer/pitch.py:104-145, 565-570)
BALLS_AND_BOXES = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="3 3 .1"/>
    <geom name="block" type="box" size=".3 .2 .1" pos="0 0 .1" euler="0 0 25"/>
    <geom name="ramp" type="box" size=".4 .3 .02" pos=".9 0 .15" euler="0 20 0"/>
    <body name="a" pos="0.05 0.02 .45">
      <freejoint/>
      <geom name="a" type="sphere" size=".1" condim="6" priority="1" friction=".7 .05 .05"/>
    </body>
    <body name="b" pos=".85 .05 .5">
      <freejoint/>
      <geom name="b" type="sphere" size=".08"/>
    </body>
  </worldbody>
</mujoco>
"""

BALL_NEAR_BOX = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <geom name="box" type="box" size=".3 .2 .1" pos=".1 -.2 .3" euler="20 35 50" margin="5"/>
    <body name="a" pos="0 0 1">
      <freejoint/>
      <geom name="a" type="sphere" size="0.07" margin="5"/>
    </body>
  </worldbody>
</mujoco>
"""

GPU_MODELS = {
    'readme_box': README_BOX,
    'box_on_floor': BOX_ON_FLOOR,
    'ball_on_floor': BALL_ON_FLOOR,
    'primitives': PRIMITIVES,
    'balls_and_boxes': BALLS_AND_BOXES,
}
# (STACKED_BOXES joins this dict below, once it is defined)

# K2 with its sensor (wrapper/core_test.py:329-344): the touch site covers the cube
CUBE_WITH_TOUCH = """
<mujoco>
  <option gravity="0 0 -9.81"/>
  <worldbody>
    <geom name="floor" type="plane" pos="0 0 0" size="10 10 0.1"/>
    <body name="cube" pos="0 0 0.1">
      <geom type="box" size="0.1 0.1 0.1" mass="1"/>
      <site name="cube_site" type="box" size="0.1 0.1 0.1"/>
      <joint type="slide"/>
    </body>
  </worldbody>
  <sensor>
    <touch name="touch_sensor" site="cube_site"/>
  </sensor>
</mujoco>
"""

# ---------------------------------------------------------------------------
# Models of the reference-independent closed-form checks (SURVEY.md Appendix D,
# tests/test_closed_form.py).  Each is a suite model with the terms that would
# break the invariant switched off in the compiled arrays; the GPU twin runs
# the same models through fp64 code objects (__graft_entry__.build()).
# ---------------------------------------------------------------------------
OSCILLATOR = """
<mujoco>
  <option timestep="0.01" gravity="0 0 0"/>
  <worldbody>
    <body name="mass" pos="0 0 0">
      <joint name="x" type="slide" axis="1 0 0" stiffness="30" damping="1.5"/>
      <geom type="sphere" size="0.1" mass="2"/>
    </body>
  </worldbody>
</mujoco>
"""

# A puck on two orthogonal sliders pulled by two motors through fixed tendons
# that mix the sliders (the transmission of suite/point_mass.xml); per-joint
# damping, no contacts: every dof follows a closed-form discrete map.
TENDON_PUCK = """
<mujoco>
  <option timestep="0.01"/>
  <worldbody>
    <body name="puck" pos="0 0 0.1">
      <joint name="x" type="slide" axis="1 0 0" damping="0.7"/>
      <joint name="y" type="slide" axis="0 1 0" damping="0.2"/>
      <geom type="sphere" size="0.05" mass="1.5"/>
    </body>
  </worldbody>
  <tendon>
    <fixed name="t1"><joint joint="x" coef="0.8"/><joint joint="y" coef="-0.6"/></fixed>
    <fixed name="t2"><joint joint="x" coef="0.3"/><joint joint="y" coef="0.9"/></fixed>
  </tendon>
  <actuator>
    <motor name="a1" tendon="t1" gear="2" ctrllimited="true" ctrlrange="-1 1"/>
    <motor name="a2" tendon="t2" gear="0.5" ctrllimited="true" ctrlrange="-1 1"/>
  </actuator>
</mujoco>
"""

CAPSULE_PAIR = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <body name="a" pos="0 0 0">
      <freejoint/>
      <geom name="a" type="capsule" size="0.07 0.3" margin="5"/>
    </body>
    <body name="b" pos="0 0 1">
      <freejoint/>
      <geom name="b" type="capsule" size="0.05 0.2" margin="5"/>
    </body>
  </worldbody>
</mujoco>
"""

CAPSULE_OVER_PLANE = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 .1" margin="5"/>
    <body name="a" pos="0 0 1">
      <freejoint/>
      <geom name="a" type="capsule" size="0.07 0.3" margin="5"/>
    </body>
  </worldbody>
</mujoco>
"""


CAPSULE_NEAR_BOX = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <body name="box" pos="0.1 -0.2 0.3" euler="20 -35 50">
      <freejoint/>
      <geom name="box" type="box" size="0.25 0.15 0.35" margin="5"/>
    </body>
    <body name="cap" pos="0 0 1.2">
      <freejoint/>
      <geom name="cap" type="capsule" size="0.06 0.3" margin="5"/>
    </body>
  </worldbody>
</mujoco>
"""

BOX_NEAR_BOX = """
<mujoco>
  <option gravity="0 0 0"/>
  <worldbody>
    <body name="a" pos="0 0 0">
      <freejoint/>
      <geom name="a" type="box" size="0.3 0.2 0.25"/>
    </body>
    <body name="b" pos="0 0 1">
      <freejoint/>
      <geom name="b" type="box" size="0.15 0.25 0.1"/>
    </body>
  </worldbody>
</mujoco>
"""

# three boxes stacked on the floor (the K3 balance, core_test.py:461-484, layer by
# layer: each interface carries the weight of everything above it), and a
# capsule lying on a box
STACKED_BOXES = """
<mujoco>
  <option timestep="0.002"/>
  <worldbody>
    <geom name="floor" type="plane" size="5 5 .1"/>
    <body name="b0" pos="0 0 0.1">
      <freejoint/>
      <geom name="b0" type="box" size="0.3 0.3 0.1" mass="3"/>
    </body>
    <body name="b1" pos="0.03 -0.02 0.28">
      <freejoint/>
      <geom name="b1" type="box" size="0.2 0.22 0.08" mass="2"/>
    </body>
    <body name="b2" pos="0.01 0.02 0.42" euler="0 0 30">
      <freejoint/>
      <geom name="b2" type="box" size="0.1 0.12 0.06" mass="1"/>
    </body>
    <body name="log" pos="0.0 0.03 0.51" euler="90 0 20">
      <freejoint/>
      <geom name="log" type="capsule" size="0.03 0.05" mass="0.5"/>
    </body>
  </worldbody>
</mujoco>
"""


GPU_MODELS['stacked_boxes'] = STACKED_BOXES


def closed_form_models():
  """name -> (compiled Model, build mode of its fp64 device code object)."""
  import numpy as np
  import helpers
  from dm_control_amd.mjcf import compiler
  from dm_control_amd.mjcf import model as mdl
  out = {}

  def frictionless(m, integrator):
    """No dissipation, no actuation, no contacts or limits."""
    m.dof_damping[:] = 0
    m.jnt_stiffness[:] = 0
    m.opt.disableflags |= (mdl.DSBL_CONTACT | mdl.DSBL_LIMIT | mdl.DSBL_ACTUATION)
    m.opt.integrator = integrator
    return m

  for integ, tag, dt_scale in ((1, 'rk4', 1.0), (0, 'euler', 1.0), (0, 'euler_half', 0.5)):
    for name, key in (('cheetah', 'cheetah_chain_'), ('acrobot', 'acrobot_')):
      m = frictionless(helpers.load_model(name), integ)
      m.opt.timestep = m.opt.timestep*dt_scale
      out[key + tag] = (m, 'auto')
  # humanoid adrift: no gravity, no contacts; joint limits, damping and motors
  # are internal forces and keep the momentum
  for tag, dt_scale in (('', 1.0), ('_half', 0.5)):
    h = helpers.load_model('humanoid')
    h.opt.gravity[:] = 0
    h.opt.disableflags |= mdl.DSBL_CONTACT
    h.opt.integrator = 1
    h.opt.timestep = h.opt.timestep*dt_scale
    out['humanoid_adrift' + tag] = (h, 'coop')
  out['oscillator'] = (compiler.from_xml_string(OSCILLATOR), 'auto')
  # cart-pole pushed against its slider limit; a stiff hinge damper lets the
  # pole come to rest quickly (the limit's solref/solimp are the reference's)
  c = helpers.load_model('cartpole')
  c.dof_damping[1] = 0.5
  c.opt.integrator = 0
  out['cartpole_at_limit'] = (c, 'auto')
  out['cube_with_touch'] = (compiler.from_xml_string(CUBE_WITH_TOUCH), 'auto')
  out['tendon_puck'] = (compiler.from_xml_string(TENDON_PUCK), 'auto')
  return out
