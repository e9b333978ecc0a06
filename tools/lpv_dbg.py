"""Find LPV closed-loop steps whose interior-point solve ends at the iteration cap and dump the inputs of that step."""
import numpy as np, sys, torch
sys.path.insert(0, '.')
from mpc_amd.batched import SkidpadBatch
B, N, Tf = 4096, 80, 0.8
rng = np.random.default_rng(1000)
py = rng.uniform(-0.3, 0.3, B); psi = rng.uniform(-0.1, 0.1, B)
vy = rng.uniform(-0.2, 0.2, B); r = rng.uniform(-0.2, 0.2, B); dl = rng.uniform(-0.05, 0.05, B)
st = np.stack([np.full(B, 15.0), py, np.cos(psi), np.sin(psi), np.full(B, 8.0), vy, r, dl], axis=1)
st[:, 4] = np.random.default_rng(1).uniform(6.0, 12.0, B)
sim = SkidpadBatch(N, Tf, st, model="LPV")
dumped = 0
hist = np.zeros(90, int)
for t in range(110):
    pre = (sim.state.clone(), sim.lap.clone(), sim.prev_progress.clone(), sim.lin_x.clone(), sim.lin_u.clone())
    sim.step()
    stt = sim.status.cpu().numpy(); it = sim.iters.cpu().numpy()
    hist += np.bincount(np.minimum(it, 89), minlength=90)
    bad = np.nonzero(stt)[0]
    for b in bad[:2]:
        if dumped < 6:
            np.savez(f"gpurun_out/lpv_bad_{dumped}.npz", state=pre[0][b].cpu().numpy(), lap=pre[1][b].cpu().numpy(), pp=pre[2][b].cpu().numpy(),
                     lin_x=pre[3][b].cpu().numpy(), lin_u=pre[4][b].cpu().numpy(), u0=sim.u0[b].cpu().numpy(), t=t, b=b)
            print("bad", t, b, stt[b], it[b], pre[0][b].cpu().numpy().tolist()); dumped += 1
print("iters hist", hist.tolist())
