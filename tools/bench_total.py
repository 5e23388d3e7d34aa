"""total_loss of SfMLearner_uni.py:40-176 end to end from the raw network outputs (disparity pyramid at H/2^s,
translation pyramid, poses) through the Python mirror: normalise -> upsample -> fused warp/photometric loss ->
smoothness -> all adjoints.  c2 shapes; prints ms per call and the per-kernel breakdown of the pyramid kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd
from sfm_amd import synth
B, H, W, S, N = 8, 128, 416, 4, 2
dev = torch.device("cuda:0")
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, full_res=False), dev)
raw = [(d * 3.0).requires_grad_(True) for d in inp["disp"]]
trans = [t.requires_grad_(True) for t in inp["trans"]]
pose = inp["pose"].requires_grad_(True)
def step():
    for t in raw + trans + [pose]:
        t.grad = None  # optimizer.zero_grad(set_to_none=True)
    dn, dp = sfm_amd.normalize_disp(raw)
    loss = sfm_amd.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], dp, pose, trans, pred_disp=dn, smooth_weight=0.01)
    loss.backward()
    return loss
for _ in range(10): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("N_STEPS", "100"))
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"total_loss fwd+bwd from raw pyramids, c2: {dt*1e3:.3f} ms/call  loss {float(l.detach()):.5f}  "
      f"({B*H*W*S*N*2/dt/1e9:.2f} Gpix/s incl. pyramid + smoothness + host overhead)")
