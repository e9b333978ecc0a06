"""LPVMPC.LPVOcp of the reference, backed by libmpc_amd (see mpc_amd.controllers)."""
from .controllers import LPVOcp  # noqa: F401
