"""NLOcp of the reference (NLMPC.py): nonlinear 7-state MPC.  Not built in this round."""


class NLOcp:
    def __init__(self, N, Tf, debug=False, **kw):
        raise NotImplementedError("the nonlinear SQP controller (NLMPC.py) is not implemented yet")
