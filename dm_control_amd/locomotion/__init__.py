"""Locomotion models on the batched step (stage 1).

Only the PHYSICS prerequisites of `dm_control.locomotion.soccer` with humanoid
walkers live here: frozen MJCF documents of the position-controlled CMU humanoid
(walkers/cmu_humanoid.py:183-428), the soccer ball (soccer/soccer_ball.py:88-96)
and a fixed-size pitch, compiled by the in-tree compiler and stepped by the
same kernels as the suite.  The composer machinery around them (PyMJCF,
observables, per-episode recompilation, the mocap initialiser) is not built.
"""
