"""Stand-alone view-synthesis operators at c2 size (B=8 416x128, one source): projective_warp / projective_inverse_warp forward + backward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth, utils as U
dev = torch.device("cuda:0")
B, H, W = 8, 128, 416
inp = synth.to_device(synth.make_inputs(B, H, W, S=1, N=2, seed=synth.SEED), dev)
img = inp["tgt"]; depth = inp["depth"][0].squeeze(3).contiguous(); pose = inp["pose"][:, 0].contiguous(); K = inp["K"]
trans = inp["trans"][0][..., :3].contiguous()
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, fn in (("projective_warp", U.projective_warp), ("projective_inverse_warp", U.projective_inverse_warp)):
    d = depth.clone().requires_grad_(True); im = img.clone().requires_grad_(True); po = pose.clone().requires_grad_(True)
    t = trans.clone().requires_grad_(True)
    out = fn(im, d, po, K, False, True, t)
    gy = torch.randn_like(out[0])
    def both():
        o = fn(im, d, po, K, False, True, t)
        torch.autograd.grad(o[0], (im, d, po, t), gy)
    with torch.no_grad():
        tf = timed(lambda: fn(img, depth, pose, K, False, True, trans))
    print(f"{name}: fwd {tf:.1f} us   fwd+bwd {timed(both):.1f} us   ({B * H * W / 1e6:.2f} Mpix; through the Python mirror, allocations included)")
