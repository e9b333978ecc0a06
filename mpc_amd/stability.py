"""Terminal-set helpers with the function names of the reference's stability.py.  The vertex test of
check_control_amissible_invariance runs on the GPU for whole batches of level values, and so do the linear programmes of
estimate_control_amissible_invariant_set (cvxpy / ECOS in the reference, stability.py:7-90): the 2 (nx + nu) programmes the
loop asks per power of the closed-loop matrix are one launch (mpc_lp_maximize, one wavefront per programme)."""
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


def lp_maximize_batched(A, b, C, device=None):
    """max c'x s.t. A x <= b for every row c of C (nlp, d), on the GPU; returns (optima (nlp,), maximisers (nlp, d),
    status (nlp,): 0 optimum, 1 unbounded, 2 not finished)."""
    dev = _dev.require_gpu(device)
    A = np.ascontiguousarray(A, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    C = np.ascontiguousarray(np.atleast_2d(C), dtype=np.float64)
    m, d = A.shape
    if b.shape[0] != m or C.shape[1] != d:
        raise ValueError("A (m, d), b (m,), C (nlp, d)")
    Ad, bd, Cd = _dev.f64(A, dev), _dev.f64(b, dev), _dev.f64(C, dev)
    opt = torch.empty(C.shape[0], dtype=torch.float64, device=dev)
    xs = torch.empty((C.shape[0], d), dtype=torch.float64, device=dev)
    st = torch.empty(C.shape[0], dtype=torch.int32, device=dev)
    rc = _lib.load().mpc_lp_maximize(_dev.stream_ptr(dev), int(C.shape[0]), int(d), int(m), _dev.ptr(Ad), _dev.ptr(bd), _dev.ptr(Cd),
                                     _dev.ptr(opt), _dev.ptr(xs), _dev.ptr(st))
    _lib.check(rc, None, "mpc_lp_maximize")
    return opt.cpu().numpy(), xs.cpu().numpy(), st.cpu().numpy()


def estimate_control_amissible_invariant_set(K, A, B, ubx, ubu, verbose=False, n_limit=300, save_path=None, device=None):
    """stability.py:7-90.  Grows the set { x : |phi^k x| <= ubx, |K phi^k x| <= ubu, k = 1..n-1 } (phi = A - B K) one power
    at a time; after each step 2 (len(ubx) + len(ubu)) linear programmes maximise the components of phi^n x / ubx and
    K phi^n x / ubu (both signs) over the set -- one GPU launch for all of them -- and the loop ends when every optimum is
    below 5e-2 (the reference's test) or n exceeds `n_limit` (the reference's 1000 would need 12000 rows; the kernel
    takes 4096); the rows are then rebuilt by reconstruct_control_amissible_invariant_set(n, ...).
    Kept as the reference has them: the first power constrained is phi^1, not phi^0 (x itself is never bounded); an
    unbounded programme counts as +inf and keeps the loop going; where the reference catches cvxpy's SolverError and
    skips the rest of that group of programmes, a programme the kernel does not finish skips the rest of its group the
    same way.  The optimum of each programme is an interior-point optimum like ECOS's (1e-10 relative here): the stopping
    index n can differ by one step where an optimum sits within solver tolerance of 5e-2.  Returns the rows C (C x <= 1)."""
    K = np.atleast_2d(np.asarray(K, dtype=np.float64)); A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    ubx = np.asarray(ubx, dtype=np.float64).reshape(-1); ubu = np.asarray(ubu, dtype=np.float64).reshape(-1)
    phi = A - B @ K
    phi_np1 = phi.copy()
    rows, rhs = [], []
    n = 1
    while True:
        M = phi_np1
        rows += [M, -M, K @ M, -(K @ M)]
        rhs += [ubx, ubx, ubu, ubu]
        n += 1
        phi_np1 = phi_np1 @ phi
        groups = [[phi_np1[j] / ubx[j] for j in range(len(ubx))], [(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))],
                  [-phi_np1[j] / ubx[j] for j in range(len(ubx))], [-(K @ phi_np1)[j] / ubu[j] for j in range(len(ubu))]]
        opt, _, status = lp_maximize_batched(np.concatenate(rows), np.concatenate(rhs), np.array([c for g in groups for c in g]), device)
        opt_values, pos = [], 0
        for group in groups:
            for k in range(len(group)):
                if status[pos + k] == 2:
                    if verbose:
                        print(f"n = {n}: programme not finished; rest of its group skipped")
                    break
                opt_values.append(float("inf") if status[pos + k] == 1 else float(opt[pos + k]))
            pos += len(group)
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
