"""OFLMPC2.OFLOcp of the reference, backed by libmpc_amd (see mpc_amd.controllers)."""
from .controllers import OFLOcp  # noqa: F401
