"""Find closed-loop steps whose interior-point solve ends at the iteration cap and dump their QP data."""
import numpy as np, sys, torch
sys.path.insert(0, '.')
from mpc_amd.batched import StepOFLBatch
B, N, Tf = 2048, 50, 0.5
for tmu in (1e-13, 1e-15):
    rng = np.random.default_rng(2)
    true0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, -10.0]), (B, 1))
    true0[:, 9] = rng.uniform(-12.0, -8.0, B)
    est0 = np.tile(np.array([-5.0, 0.0, 1.0, 0.0, 15.0, 0.0, 0.0, 0.0, 0.0, 0.0]), (B, 1))
    est0[:, 1] += rng.normal(0, 0.05, B); est0[:, 5] += rng.normal(0, 0.02, B)
    sim = StepOFLBatch(N, Tf, true0, est0, tol_mu=tmu)
    gen = torch.Generator(device=sim.device); gen.manual_seed(3)
    sigma = torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.01, 0.01], dtype=torch.float64, device=sim.device)
    noise = torch.randn((120, B, 7), dtype=torch.float64, device=sim.device, generator=gen) * sigma
    dumped = 0
    tot_bad = 0; hist = np.zeros(70, int)
    for t in range(120):
        est_before = sim.estimate.clone()
        sim.run(1, meas_noise=noise[t:t + 1])
        st = sim.status.cpu().numpy(); it = sim.iters.cpu().numpy()
        hist += np.bincount(np.minimum(it, 69), minlength=70)
        bad = np.nonzero(st)[0]
        tot_bad += len(bad)
        if len(bad) and dumped < 4 and tmu == 1e-15:
            for b in bad[:2]:
                np.save(f"gpurun_out/bad_est_{dumped}.npy", est_before[b].cpu().numpy()); dumped += 1
                print("bad", t, b, st[b], it[b], est_before[b].cpu().numpy().tolist())
    print("tol_mu", tmu, "bad total", tot_bad, "of", 120 * B, "iters hist", hist.tolist())
