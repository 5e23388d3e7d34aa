"""host-side profile (cProfile) of one eager total_loss step from raw pyramids: where the Python time goes"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfm_amd
from sfm_amd import synth
import bench
dev = torch.device("cuda:0")
B, H, W, S, N = bench.B, bench.H, bench.W, bench.S, bench.N
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, full_res=False), dev)
raw = [(d * 3.0).requires_grad_(True) for d in inp["disp"]]
trans = [t.requires_grad_(True) for t in inp["trans"]]
pose = inp["pose"].requires_grad_(True)


def step():
    for t in raw + trans + [pose]:
        t.grad = None
    dn, dp = sfm_amd.normalize_disp(raw)
    loss = sfm_amd.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], dp, pose, trans, pred_disp=dn, smooth_weight=0.01)
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()          # host time only: no synchronisation inside the loop
torch.cuda.synchronize()
print(f"host enqueue time per step {(t1 - t0) / 200 * 1e3:.3f} ms, with the GPU drained {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
