"""The checker's restatement of the nonlinear controller (NLMPC.py), certified on the CPU without the library:
its solutions must be local minimisers of the programme NLMPC.py poses, evaluated here by an independent numpy
roll-out of the model.  The reference's own solve runs inside acados (absent here) and no fixture of the reference
records an NL solution, so this certificate is what pins the checker for this controller: parity unpinned otherwise."""
import numpy as np
import pytest


def _nl_f(x, u, vx, mp, eps=0.1):
    m, Iz, lf, lr, Cf, Cr = mp                                   # NLMPC.py:108-134
    dvy = -(Cf + Cr) / (m * vx + eps) * x[4] + (-vx + (Cr * lr - Cf * lf)) / (m * vx + eps) * x[5] - Cf / m * x[6]
    dr = (lr * Cr - lf * Cf) / (Iz * vx + eps) * x[4] - (lf * lf * Cf + lr * lr * Cr) / (Iz * vx + eps) * x[5] \
        - lf * Cf / Iz * x[6]
    return np.array([vx * x[2] - x[4] * x[3], vx * x[3] + x[4] * x[2], -x[3] * x[5], x[2] * x[5], dvy, dr, u])


def _rollout(U, x0, p, dt, mp):
    X = [np.array(x0, dtype=np.float64)]
    for k, u in enumerate(U):
        x = X[-1]; vx = p[k]
        k1 = _nl_f(x, u, vx, mp); k2 = _nl_f(x + dt / 2 * k1, u, vx, mp)
        k3 = _nl_f(x + dt / 2 * k2, u, vx, mp); k4 = _nl_f(x + dt * k3, u, vx, mp)
        X.append(x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4))
    return np.array(X)


def _cost(U, x0, wp, p, dt, mp):
    Q = np.array([0, 1e5, 1e-10, 1e0, 0, 1e0, 1e0]); R = 10.0   # NLMPC.py:174-179; stage cost scaled by dt, terminal not
    X = _rollout(U, x0, p, dt, mp)
    ref = np.zeros_like(X); ref[:, :4] = wp
    e = X - ref
    return 0.5 * dt * (np.sum(Q * e[:-1] ** 2) + R * np.sum(np.asarray(U) ** 2)) + 0.5 * np.sum(100 * Q * e[-1] ** 2), X


@pytest.mark.parametrize("start", [[15.0, 0.1, 0.0, 0.0, 0.0, 0.0], [12.0, 0.0, 0.0, 0.0, 0.0, 0.0],
                                   [15.0, 0.3, 0.05, 0.1, -0.1, 0.28], [15.0, -0.25, -0.08, -0.2, 0.2, -0.04]])
def test_nl_checker_solution_is_a_local_minimiser(orc80, orc64, start):
    from oracle import oracle as O
    N, Tf = 50, 0.5
    dt = Tf / N
    cfg = O.make_cfg("NL", N, Tf)
    mp = cfg[3:9]
    px, py, psi, vy, r, delta = start
    wp, sp, _, _ = orc80.skidpad_waypoints(px, py, psi, 0, 0.0, 8.0, N, dt)
    x0_7 = np.array([px, py, np.cos(psi), np.sin(psi), vy, r, delta])
    st, traj, U, lam, info = orc80.nl_optimize(cfg, x0_7, wp, sp)
    assert st == 0 and info[2] <= 1e-11 and info[3] <= 1e-12, info
    st64, traj64, U64, _, info64 = orc64.nl_optimize(cfg, x0_7, wp, sp)
    assert st64 == 0
    np.testing.assert_allclose(U64, U, rtol=0, atol=2e-8)
    xs = np.array([0, 0, 1, 0, vy, r, delta])                   # NLMPC.py:216
    c0, X = _cost(U, xs, wp, sp, dt, mp)
    np.testing.assert_allclose(traj, X, rtol=0, atol=1e-10)     # the returned states are the model's own roll-out
    assert np.abs(U).max() <= 0.9 + 1e-12 and np.abs(X[1:N, 6]).max() <= 0.3 + 1e-12
    # feasible perturbations never lower the cost (first order: c0 is a KKT point; second order: GN curvature > 0)
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(200):
        d = rng.normal(0, 1, N) * rng.choice([1e-3, 1e-4, 1e-5])
        V = np.clip(U + d, -0.9, 0.9)
        c1, X1 = _cost(V, xs, wp, sp, dt, mp)
        if np.abs(X1[1:N, 6]).max() > 0.3:
            continue
        worst = min(worst, (c1 - c0) / max(c0, 1e-12))
    assert worst >= -1e-12, worst
    # where no bound is active the gradient itself vanishes
    if np.abs(lam).max() < 1e-12:
        g = np.array([(_cost(U + 1e-6 * e, xs, wp, sp, dt, mp)[0] - _cost(U - 1e-6 * e, xs, wp, sp, dt, mp)[0]) / 2e-6
                      for e in np.eye(N)])
        assert np.abs(g).max() <= 1e-6 * max(1.0, c0)


def test_nl_checker_keeps_the_iterate_between_calls(orc64):
    """A second call from the previous solution needs fewer SQP iterations than the cold start and lands on the same
    minimiser as a cold solve of the same problem."""
    from oracle import oracle as O
    N, Tf = 50, 0.5
    cfg = O.make_cfg("NL", N, Tf)
    wp, sp, _, _ = orc64.skidpad_waypoints(15.0, 0.2, 0.03, 0, 0.0, 8.0, N, Tf / N)
    x0 = np.array([15.0, 0.2, np.cos(0.03), np.sin(0.03), 0.05, -0.05, 0.01])
    st, traj, U, lam, info = orc64.nl_optimize(cfg, x0, wp, sp)
    x1 = orc64.plant_rk4(np.array([15.0, 0.2, np.cos(0.03), np.sin(0.03), 8.0, 0.05, -0.05, 0.01]), U[0], Tf / N, cfg[3:9])
    wp1, sp1, _, _ = orc64.skidpad_waypoints(x1[0], x1[1], np.arctan2(x1[3], x1[2]), 0, 0.0, 8.0, N, Tf / N)
    red = x1[[0, 1, 2, 3, 5, 6, 7]]
    sw, _, Uw, _, iw = orc64.nl_optimize(cfg, red, wp1, sp1, U)
    sc, _, Uc, _, ic = orc64.nl_optimize(cfg, red, wp1, sp1)
    assert sw == 0 and sc == 0 and iw[1] <= ic[1]
    np.testing.assert_allclose(Uw, Uc, rtol=0, atol=1e-8)
