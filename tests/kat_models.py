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

GPU_MODELS = {
    'readme_box': README_BOX,
    'box_on_floor': BOX_ON_FLOOR,
    'ball_on_floor': BALL_ON_FLOOR,
    'primitives': PRIMITIVES,
}
