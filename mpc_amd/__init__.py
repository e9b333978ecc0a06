"""mpc_amd: MI355X-native per-step control path of the Formula-Student skidpad MPC stack.

Same class surface as the reference (LOcp / LPVOcp / OFLOcp, SkidpadSimulator, StepSimulator, Dynamics,
CarEKF, SkidpadPlanner, StepPlanner); the arithmetic runs in libmpc_amd.so (HIP, gfx950)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
