"""NLOcp of the reference (NLMPC.py): nonlinear 7-state MPC."""
from .controllers import NLOcp  # noqa: F401
