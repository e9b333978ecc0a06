"""LMPC2.LOcp of the reference, backed by libmpc_amd (see mpc_amd.controllers)."""
from .controllers import LOcp  # noqa: F401
