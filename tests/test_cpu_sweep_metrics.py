"""Step-response metrics and run-time statistics of plot_generator.py:1057-1072, 1160-1221 as restated in
mpc_amd/sweeps.py, on series whose metrics are known by construction (plot_generator itself imports the acados
controllers at module level and cannot be imported here), and the parameter sets of the sweep points."""
import numpy as np


def test_performance_metrics_known_series():
    from mpc_amd import sweeps
    n, dt = 300, 0.01
    t = np.arange(n) * dt
    y = 1 - np.exp(-t / 0.2) * np.cos(12 * t)                 # overshooting step response settling at ~1
    m = sweeps.performance_metrics(y, 2)
    first = int(np.argmax(y >= 0.9 * y[-1]))
    assert m["State"] == "x2" and m["Rise Time (s)"] == first * dt
    outside = np.nonzero(np.abs(y - y[-1]) > 0.02 * y[-1])[0]
    assert m["Settling Time (s)"] == (n - (n - 1 - outside[-1])) * dt     # reversed index of the last sample outside
    assert m["Overshoot"] == y.max() - y[-1]
    # quirk kept from the reference: a series that never leaves the band gets None, and so does one whose last-but-one
    # search index is 0
    flat = np.ones(n)
    assert sweeps.performance_metrics(flat, 0)["Settling Time (s)"] is None
    df = sweeps.compute_performance_metrics([np.stack([y, flat], axis=1)] * 2, ["a", "b"])
    assert list(df["State"]) == ["x1", "x1", "x2", "x2"] and list(df["Model"]) == ["a", "b", "a", "b"]
    assert df["Rise Time (s)"][0] == first * dt


def test_compute_time_metrics():
    from mpc_amd import sweeps
    rng = np.random.default_rng(0)
    a, b = rng.uniform(0.1, 2.0, 57), rng.uniform(0.1, 2.0, 300)
    df = sweeps.compute_time_metrics([a, b])
    assert list(df["Dataset"]) == ["Dataset 1", "Dataset 2"]
    np.testing.assert_allclose(df["Mean Runtime (ms)"], [a.mean(), b.mean()])
    np.testing.assert_allclose(df["Median Runtime (ms)"], [np.median(a), np.median(b)])
    np.testing.assert_allclose(df["Standard Deviation (ms)"], [a.std(), b.std()])
    np.testing.assert_allclose(df["90th Percentile Runtime (ms)"], [np.percentile(a, 90), np.percentile(b, 90)])
    np.testing.assert_allclose(df["Minimum Runtime (ms)"], [a.min(), b.min()])
    np.testing.assert_allclose(df["Maximum Runtime (ms)"], [a.max(), b.max()])


def test_sweep_points_and_grids():
    from mpc_amd import sweeps
    # 10000 // 0.1 is 99999.0 in floating point, so the reference's q_y grid has 4 points, not 5 (plot_generator.py:169-185)
    assert sweeps.decade_values(0.1, 10000) == [0.1 * 10 ** i for i in range(4)]
    assert sweeps.decade_values(1, 10000) == [1, 10, 100, 1000]                         # :253-269
    assert len(sweeps.decade_values(0.01, 10000)) == 5                                  # :414-431, same floor-division effect
    p = sweeps.sweep_parameters("LPV", "q_y", 100.0)
    assert p["controller"]["q"] == 1 and p["controller"]["Q"][1][1] == 100.0
    base = sweeps.sweep_parameters("L", "r", 3.0)
    assert base["controller"]["r"] == 3.0
    from mpc_amd.controllers import read_parameters
    assert read_parameters("parameters_L.yaml")["controller"]["r"] != 3.0               # the packaged file is not touched
    ref = sweeps.step_reference(np.array([[-4.0, 0, 1, 0, 15, 0, 0, 0], [0.5, 0, 1, 0, 15, 0, 0, 0], [1, 0, 1, 0, 15, 0, 0, 0]]), -5.0)
    np.testing.assert_array_equal(ref[:, 0], [-5.0, -4.0, 0.5]); np.testing.assert_array_equal(ref[:, 1], [0, 0, 1])


def test_reconstructed_invariant_set_rows():
    """reconstruct_control_amissible_invariant_set (stability.py:93-109): host arithmetic, equal to the restatement, and
    its rows are the constraints |phi^k x| <= ubx, |K phi^k x| <= ubu written as C x <= 1."""
    import numpy as np
    from oracle import oracle as O
    from mpc_amd.stability import reconstruct_control_amissible_invariant_set
    rng = np.random.default_rng(2)
    A = np.eye(5) + 0.05 * rng.normal(size=(5, 5)); B = rng.normal(size=(5, 1)); K = rng.normal(size=(1, 5)) * 0.3
    ubx = np.array([100, 100, 100, 100, 0.3]); ubu = np.array([0.9])
    n = 6
    C = reconstruct_control_amissible_invariant_set(n, K, A, B, ubx, ubu)
    np.testing.assert_array_equal(C, O.reconstruct_invariant_set(n, K, A, B, ubx, ubu))
    assert C.shape == (12 * (n + 1), 5)
    phi = A - B @ K
    x = rng.normal(size=5) * 0.05
    for k in range(n + 1):
        xk = np.linalg.matrix_power(phi, k) @ x
        blk = C[12 * k: 12 * (k + 1)] @ x
        # one 12-row block per k: rows 0-4 / 5-9 carry x_k / ubx (both signs), rows 10 / 11 carry K x_k / ubu (both signs)
        np.testing.assert_allclose(blk[10], (K @ xk)[0] / ubu[0], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(blk[11], -(K @ xk)[0] / ubu[0], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(blk[:5], xk / ubx, rtol=1e-12, atol=1e-14)
