"""The benchmark workloads at their full batch sizes, through properties that do not need the checker at that size:
every scenario is its own closed loop (skidpad_simulator.py:109 runs exactly one), so a scenario advanced inside the full
batch must equal the same scenario advanced in a small batch bit for bit, whatever the launch's scenario order and
residency; inputs and steering angles stay inside their bounds; and a few scenarios of the full batch are followed with
the checker's loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _utol(u_ref, umax=0.9):
    return 10 * 1e-5 * np.maximum(np.abs(u_ref), 1e-3 * umax)       # closed loop: differences feed back


def _build(name, st, mp=None):
    import bench
    from mpc_amd.batched import SkidpadBatch
    w = bench.WORKLOADS[name]
    kw = dict(sqp_tol=1e-30, sqp_max_iter=3) if name == "c5" else {}
    if name == "c3":
        kw["solver_layout"] = "lane"      # what the full batch gets by itself; the small batch must run the same solver
    sim = SkidpadBatch(w["N"], w["Tf"], st, model=w["kind"], **kw)
    if mp is not None:
        sim.ocp.set_model_params(mp)
    return sim


# step counts: the workloads' own run lengths (c2 50 + 1000, c3 500, c5 200)
@pytest.mark.parametrize("name,nsteps,nsub,nref,ref_steps", [("c2", 1050, 64, 8, 200), ("c3", 500, 32, 3, 25),
                                                               ("c5", 200, 64, 2, 10)])
def test_full_batch_scenarios_are_independent_and_follow_checker(orc64, name, nsteps, nsub, nref, ref_steps):
    import torch
    import bench
    from oracle import oracle as O
    w = bench.WORKLOADS[name]
    B, N, Tf, kind = w["B"], w["N"], w["Tf"], w["kind"]
    rng = np.random.default_rng(99)
    st = bench.initial_states(B, 0)
    mp = None
    if name == "c3":
        st[:, 4] = rng.uniform(6.0, 12.0, B)
    full = _build(name, st)
    if name == "c5":                                                  # tyre / mass sweep: one parameter set per scenario
        mp = np.tile(full.ocp.model_params(), (B, 1))
        mp[:, 0] = rng.uniform(170, 230, B); mp[:, 1] = rng.uniform(280, 310, B)
        mp[:, 4] *= rng.uniform(0.6, 1.2, B); mp[:, 5] *= rng.uniform(0.6, 1.2, B)
        full.ocp.set_model_params(mp)
    xs, us = full.run(nsteps, record=True)
    idx = np.sort(rng.choice(B, nsub, replace=False))
    sub = _build(name, st[idx], None if mp is None else mp[idx])
    xs_s, us_s = sub.run(nsteps, record=True)
    torch.cuda.synchronize()
    tidx = torch.as_tensor(idx, device=us.device)
    assert torch.equal(us[:, tidx], us_s)
    assert torch.equal(xs[:, tidx], xs_s)
    assert torch.equal(full.iters[tidx], sub.iters) and torch.equal(full.lap[tidx], sub.lap)
    # bounds of the OCP hold on what was applied (LMPC2.py:196-210); the plant integrates delta' = u exactly
    assert float(us.abs().max()) <= 0.9 + 1e-9
    assert float(xs[:, :, 7].abs().max()) <= 0.3 + 1e-6
    assert bool(torch.isfinite(xs).all())
    status = full.status.cpu().numpy()
    if name == "c5":
        assert set(np.unique(status)) <= {0, 1}                       # 1 = the fixed three SQP iterations were used up
    else:
        assert (status == 0).all(), np.bincount(status)                 # every solve of every step ended at its minimiser
    # a few scenarios of the full batch against the checker's loop
    us_h = us[:ref_steps, tidx[:nref]].cpu().numpy()
    for j in range(nref):
        b = idx[j]
        cfg = O.make_cfg(kind, N, Tf, sqp_tol=1e-30, sqp_max_iter=3) if kind == "NL" else O.make_cfg(kind, N, Tf)
        if mp is not None:
            cfg[3:9] = mp[b]
        ref = st[b:b + 1].copy(); laps = np.zeros(1, np.int32); pp = np.zeros(1)
        uh, _ = orc64.skidpad_run(cfg, ref, laps, pp, float(st[b, 4]), ref_steps)
        err = np.abs(us_h[:, j] - uh[:, 0])
        assert (err <= _utol(uh[:, 0])).all(), (name, b, int(np.argmax(err / _utol(uh[:, 0]))), err.max())


def test_full_batch_output_feedback_scenarios_are_independent():
    """The output-feedback workload (OFL controller on the EKF estimate, noisy measurements) at its full batch size:
    scenarios inside the full batch equal the same scenarios, with the same noise samples, advanced in a small batch."""
    import torch
    import bench
    from mpc_amd.batched import StepOFLBatch
    w = bench.WORKLOADS["c4"]
    B, N, Tf, nsteps, nsub = w["B"], w["N"], w["Tf"], 300, 48
    rng = np.random.default_rng(5)
    true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))   # plot_generator.py:18-29
    true0[:, 9] = rng.uniform(-12.0, -8.0, B)
    est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))      # simulator.py:214-224
    est0[:, 1] += rng.normal(0, 0.05, B); est0[:, 5] += rng.normal(0, 0.05, B)
    sigma = np.array([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01])                                        # continuous_dynamics.py:68-72
    noise = rng.standard_normal((nsteps, B, 7)) * sigma
    full = StepOFLBatch(N, Tf, true0, estimates=est0)
    xs, us, es = full.run(nsteps, record=True, meas_noise=noise)
    idx = np.sort(rng.choice(B, nsub, replace=False))
    sub = StepOFLBatch(N, Tf, true0[idx], estimates=est0[idx])
    xs_s, us_s, es_s = sub.run(nsteps, record=True, meas_noise=np.ascontiguousarray(noise[:, idx]))
    torch.cuda.synchronize()
    tidx = torch.as_tensor(idx, device=us.device)
    assert torch.equal(us[:, tidx], us_s) and torch.equal(xs[:, tidx], xs_s) and torch.equal(es[:, tidx], es_s)
    assert float(us.abs().max()) <= 0.9 + 1e-9
    assert bool(torch.isfinite(xs).all()) and bool(torch.isfinite(es).all())
    # status 4 marks steps whose estimated steering angle lay outside the bound (the bound is widened there); no other flag
    assert set(np.unique(full.status.cpu().numpy())) <= {0, 4}
    # eight scenarios of the full batch followed by the checker's loop (simulator.py:206-281) through all 300 steps, with
    # the same noise samples: inputs at the closed-loop bar, plant states and estimates to 1e-6
    from oracle import oracle as O
    orc = O.Oracle("f64")
    pick = idx[:8]
    cfg = O.make_cfg("OFL", N, Tf)
    x = true0[pick].copy(); e = est0[pick].copy(); P = np.tile(np.diag(O.EKF_P0_DIAG), (8, 1, 1))
    u_ref, x_ref, e_ref, st_ref, _ = orc.ofl_run(cfg, x, e, P, 15.0, nsteps, noise=np.ascontiguousarray(noise[:, pick]))
    u_gpu = us[:, tidx[:8]].cpu().numpy(); x_gpu = xs[:, tidx[:8]].cpu().numpy(); e_gpu = es[:, tidx[:8]].cpu().numpy()
    err = np.abs(u_gpu - u_ref)
    assert (err <= _utol(u_ref)).all(), (np.unravel_index(np.argmax(err / _utol(u_ref)), err.shape), err.max())
    np.testing.assert_allclose(x_gpu, x_ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(e_gpu, e_ref, rtol=0, atol=1e-6)
    assert (full.status.cpu().numpy()[pick] == st_ref).all()
