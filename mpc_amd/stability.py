"""Terminal-set helpers with the function names of the reference's stability.py.  The vertex test of
check_control_amissible_invariance runs on the GPU for whole batches of level values; the LP loop of
estimate_control_amissible_invariant_set (cvxpy / ECOS in the reference, stability.py:7-90) is set-up arithmetic on the
host, as in the reference, with scipy's LP solver in place of ECOS."""
from __future__ import annotations

import numpy as np
import torch

from . import _dev, _lib


class ClosedLoopMap:
    """The map the reference hands to the checks as a Python callable `f`: x+ = A x + B sat(-K x, +-umax)
    (umax = inf: the linear law x+ = (A - B K) x).  Here it is data, so that the GPU can evaluate it."""

    def __init__(self, A, B, K, umax=np.inf):
        self.A = np.ascontiguousarray(A, dtype=np.float64)
        d = self.A.shape[0]
        if self.A.shape != (d, d) or not 1 <= d <= 6:
            raise ValueError("A must be square with 1 <= d <= 6")
        self.B = np.ascontiguousarray(np.asarray(B, dtype=np.float64).reshape(d))
        self.K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(d))
        self.umax = float(umax)
        self.d = d

    def __call__(self, x):
        """Host evaluation for callers that want the map itself (not used by the checks)."""
        x = np.asarray(x, dtype=np.float64)
        return self.A @ x + self.B * np.clip(-self.K @ x, -self.umax, self.umax)


def reconstruct_control_amissible_invariant_set(n, K, A, B, ubx, ubu, save_path=None):
    """stability.py:93-112: rows C x <= 1 of the set { x : |phi^k x| <= ubx, |K phi^k x| <= ubu, k = 0..n }, stacked as
    [F + G K; (F + G K) phi; ...; (F + G K) phi^n].  Set-up arithmetic on the host, like the reference; the plot is not
    reproduced and the array is only written to disk when save_path is given (the reference always writes
    control_admissible_invariant_set.npy into the working directory)."""
    K = np.atleast_2d(np.asarray(K, dtype=np.float64)); A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    ubx = np.asarray(ubx, dtype=np.float64); ubu = np.asarray(ubu, dtype=np.float64)
    phi = A - B @ K
    F = np.concatenate((np.diag(1 / ubx), np.diag(-1 / ubx), np.zeros((2 * len(ubu), len(ubx)))), axis=0)
    G = np.concatenate((np.zeros((2 * len(ubx), len(ubu))), np.diag(1 / ubu), np.diag(-1 / ubu)), axis=0)
    C_initial = F + G @ K
    Cm = C_initial
    for _ in range(n):
        C_initial = C_initial @ phi
        Cm = np.concatenate((Cm, C_initial), axis=0)
    if save_path is not None:
        np.save(save_path, Cm)
    return Cm


def estimate_control_amissible_invariant_set(K, A, B, ubx, ubu, verbose=False, n_limit=1000, save_path=None):
    """stability.py:7-90.  Grows the set { x : |phi^k x| <= ubx, |K phi^k x| <= ubu, k = 1..n-1 } (phi = A - B K) one power
    at a time; after each step 2 (len(ubx) + len(ubu)) linear programmes maximise the components of phi^n x / ubx and
    K phi^n x / ubu (both signs) over the set, and the loop ends when every optimum is below 5e-2 (the reference's test)
    or n exceeds `n_limit`; the rows are then rebuilt by reconstruct_control_amissible_invariant_set(n, ...).
    Kept as the reference has them: the first power constrained is phi^1, not phi^0 (x itself is never bounded); an
    unbounded programme counts as +inf and keeps the loop going.  Where the reference catches cvxpy's SolverError and
    skips the rest of that group of programmes, a programme HiGHS cannot finish is skipped the same way.  The optimum
    of each programme is HiGHS's instead of ECOS's (interior point, ~1e-8): the stopping index n can differ by one step
    where an optimum sits within solver tolerance of 5e-2.  Returns the row matrix C (C x <= 1)."""
    from scipy.optimize import linprog
    K = np.atleast_2d(np.asarray(K, dtype=np.float64)); A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    ubx = np.asarray(ubx, dtype=np.float64).reshape(-1); ubu = np.asarray(ubu, dtype=np.float64).reshape(-1)
    phi = A - B @ K
    phi_np1 = phi.copy()
    rows, rhs = [], []
    n = 1

    def maximise(c):
        r = linprog(-c, A_ub=np.concatenate(rows), b_ub=np.concatenate(rhs), bounds=(None, None), method="highs")
        if r.status == 0:
            return float(-r.fun)
        if r.status == 3:
            return float("inf")
        raise RuntimeError(r.message)

    while True:
        M = phi_np1
        rows += [M, -M, K @ M, -(K @ M)]
        rhs += [ubx, ubx, ubu, ubu]
        n += 1
        phi_np1 = phi_np1 @ phi
        opt_values = []
        groups = [[phi_np1[j] / ubx[j] for j in range(len(ubx))], [(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))],
                  [-phi_np1[j] / ubx[j] for j in range(len(ubx))], [-(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))]]
        for group in groups:
            try:
                for c in group:
                    opt_values.append(maximise(c))
            except RuntimeError as e:
                if verbose:
                    print(f"n = {n}: programme not solved ({e}); rest of its group skipped")
        if verbose:
            print(f"n = {n}: optima {opt_values}")
        if np.all(np.array(opt_values) < 5e-2) or n > n_limit:
            return reconstruct_control_amissible_invariant_set(n, K, A, B, ubx, ubu, save_path=save_path)


def _eigh(P):
    eigvals, eigvecs = np.linalg.eigh(np.asarray(P, dtype=np.float64))
    if not (eigvals > 0).all():
        raise ValueError("P must be positive definite")
    return np.ascontiguousarray(eigvals), np.ascontiguousarray(eigvecs)


def check_invariance_batched(P, f: ClosedLoopMap, c_values, device=None) -> np.ndarray:
    """check_control_amissible_invariance for every level in c_values, one launch; returns a bool array."""
    if not isinstance(f, ClosedLoopMap):
        raise TypeError("f must be a ClosedLoopMap (A, B, K, umax): a Python callable cannot run on the GPU, and there is "
                        "no CPU fallback")
    dev = _dev.require_gpu(device)
    eigvals, Q = _eigh(P)
    if Q.shape[0] != f.d:
        raise ValueError("P and the closed-loop map have different dimensions")
    c = _dev.f64(np.ascontiguousarray(np.atleast_1d(c_values), dtype=np.float64), dev)
    ok = torch.empty(c.shape[0], dtype=torch.int32, device=dev)
    umax = f.umax if np.isfinite(f.umax) else 1.7976931348623157e308
    rc = _lib.load().mpc_invariance_check(_dev.stream_ptr(dev), int(c.shape[0]), f.d, _lib.hptr(Q), _lib.hptr(eigvals),
                                          _lib.hptr(f.A), _lib.hptr(f.B), _lib.hptr(f.K), float(umax), _dev.ptr(c), _dev.ptr(ok))
    _lib.check(rc, None, "mpc_invariance_check")
    return ok.cpu().numpy().astype(bool)


def check_control_amissible_invariance(P, f: ClosedLoopMap, c) -> bool:
    """stability.py:155-189 for one level c."""
    return bool(check_invariance_batched(P, f, [c])[0])


def binary_search(P, f: ClosedLoopMap, c_u, epsilon=1e-1, lookahead: int = 6) -> float:
    """stability.py:192-211.  The bisection visits midpoints (c_l + c_u) / 2 one after the other; here the midpoints the
    next `lookahead` steps can reach (a binary tree of 2^lookahead - 1 levels) are evaluated in one launch and the
    reference's loop is walked through them, launch after launch, until the interval is shorter than epsilon."""
    c_u = float(c_u)
    if check_control_amissible_invariance(P, f, c_u):
        return c_u
    c_l, c_best = 0.0, 0
    while c_u - c_l > epsilon:
        levels, frontier = [], [(c_l, c_u)]
        while len(levels) < lookahead and frontier[0][1] - frontier[0][0] > epsilon:
            mids = [(lo + hi) / 2 for lo, hi in frontier]
            levels.append(mids)
            frontier = [iv for (lo, hi), m in zip(frontier, mids) for iv in ((lo, m), (m, hi))]
        ok = check_invariance_batched(P, f, np.array([m for mids in levels for m in mids]))
        pos, idx = 0, 0
        for mids in levels:
            if not c_u - c_l > epsilon:
                break
            c = mids[idx]                       # == (c_l + c_u) / 2 of the walk so far
            if ok[pos + idx]:
                c_best = c; c_l = c; idx = 2 * idx + 1
            else:
                c_u = c; idx = 2 * idx
            pos += len(mids)
    return c_best
