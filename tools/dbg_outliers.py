"""Debug aid: where do fused and oracle-fp32 per-pixel gradients differ at c2 size?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd
from sfm_amd import synth
from oracle import ref_graph as R
B, H, W, S, N = 8, 128, 416, 4, 2
dev = torch.device("cuda:0")
inp = synth.make_inputs(B, H, W, S=S, N=N, seed=41)
torch.set_num_threads(16)
depth = [d.clone().requires_grad_(True) for d in inp["depth"]]
trans = [t.clone().requires_grad_(True) for t in inp["trans"]]
pose = inp["pose"].clone().requires_grad_(True)
total, pix, sm, parts = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], depth, pose, trans, return_parts=True)
total.backward()
def run(fused):
    g = synth.to_device(inp, dev)
    dg = [d.requires_grad_(True) for d in g["depth"]]
    tg = [t.requires_grad_(True) for t in g["trans"]]
    pg = g["pose"].requires_grad_(True)
    lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], dg, pg, tg, fused=fused)
    lg.backward(); torch.cuda.synchronize()
    return lg, dg, pg, tg
lf, df, pf, tf = run(True)
lu, du, pu, tu = run(False)
print("loss fused %.8f unfused %.8f oracle %.8f" % (float(lf), float(lu), float(total)))
for s in range(S):
    es = []
    for i in range(2):
        G = parts[s * N + i]["proj_image"].detach()
        es.append(0.15 * (G - inp["tgt"]).abs() + 0.85 * R.tf_ssim(G, inp["tgt"], size=3, mean_metric=False))
    gap = ((es[0] - es[1]).abs() / torch.maximum(es[0], es[1])).amin(dim=3)   # [B,H,W] smallest relative margin over channels
    gap = -torch.nn.functional.max_pool2d(-gap[:, None], 3, 1, 1)[:, 0]
    o = depth[s].grad.squeeze(3)
    sc = float(o.abs().max())
    for name, x in (("fused", df[s].grad), ("unfused", du[s].grad)):
        e = (x.cpu().squeeze(3) - o).abs() / sc
        bad = (e > 1e-4).nonzero()
        print(f"scale {s} {name}: outliers {len(bad)} max {float(e.max()):.2e}")
        for b, y, xx in bad[:6].tolist():
            print(f"    b {b} y {y} x {xx} err {float(e[b,y,xx]):.2e} min-margin(3x3) {float(gap[b,y,xx]):.2e}")
