"""The checker's acados emulation (oracle.AcadosEmulation): what the reference's solver set-up (SQP + MIRROR
regularisation, KKT tolerance 1e-4, iterate kept between calls, LMPC2.py:262-288) makes of the posed QP, next to the exact
minimiser.  It pins nothing (acados cannot run here); it measures how far the two can be apart, so that the parity bar
"1e-5 on the inputs" is read for what it is: a bar against the minimiser of the posed problem."""
import numpy as np


def test_regularised_sqp_step_reduces_to_the_exact_solve_without_regularisation(orc64):
    from oracle import oracle as O
    N, Tf = 50, 0.5
    cfg = O.make_cfg("L", N, Tf)
    x = np.array([15.0, 0.25, np.cos(0.05), np.sin(0.05), 8.0, 0.1, -0.1, 0.02])
    wp, sp, _, _ = orc64.skidpad_waypoints(x[0], x[1], np.arctan2(x[3], x[2]), 0, 0.0, 8.0, N, Tf / N)
    x07 = x[[0, 1, 2, 3, 5, 6, 7]]
    _, traj, U, _, _ = orc64.optimize(cfg, x07, wp, sp)
    # reg_eps = 0 mirrors nothing: the step is the plain solve whatever the previous iterate
    _, traj0, U0, xs, _ = orc64.optimize_sqp_step(cfg, x07, wp, sp, np.ones((N + 1, 6)), reg_eps=0.0, tol=orc64.tol)
    np.testing.assert_allclose(U0, U, rtol=0, atol=1e-12)
    np.testing.assert_allclose(xs[:, 5], traj[:, 6], rtol=0, atol=1e-12)
    # with the regularisation and the exact solution as previous iterate the step stays where it is (proximal term zero)
    xe = np.column_stack([traj[:, 0], traj[:, 1], np.arctan2(traj[:, 3], traj[:, 2]), traj[:, 4], traj[:, 5], traj[:, 6]])
    _, _, U1, _, _ = orc64.optimize_sqp_step(cfg, x07, wp, sp, xe, reg_eps=1e-4, tol=orc64.tol)
    np.testing.assert_allclose(U1, U, rtol=0, atol=1e-9)


def test_emulated_reference_solver_stays_close_to_the_minimiser_on_the_skidpad_loop():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from report_acados_emulation import run
    r = run(steps=500, N=50, Tf=0.5, kind="L")
    assert r["sqp_iterations_histogram"][1] == 499 and r["sqp_iterations_histogram"][2] == 1   # two iterations on the first call only
    assert 1e-7 < r["max_abs_du0"] < 5e-3           # the regularisation is felt in the inputs, in their 3rd-4th digit
    assert r["max_pos_div"] < 1e-4 and r["max_yaw_div"] < 1e-4       # and not in the closed-loop trajectory
