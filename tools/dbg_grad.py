"""Debug aid: fused HIP gradients vs the oracle in fp32 and fp64, with near-tie exclusion."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd
from sfm_amd import synth
from oracle import ref_graph as R
B, H, W, S, N = 2, 128, 416, 2, 2
dev = torch.device("cuda:0")
inp = synth.make_inputs(B, H, W, S=S, N=N, seed=41)

def oracle(dt):
    c = lambda t: t.to(dt)
    depth = [c(d).clone().requires_grad_(True) for d in inp["depth"]]
    trans = [c(t).clone().requires_grad_(True) for t in inp["trans"]]
    pose = c(inp["pose"]).clone().requires_grad_(True)
    total, pix, sm, parts = R.multi_scale_loss(c(inp["tgt"]), c(inp["src"]), c(inp["K"]), depth, pose, trans, return_parts=True)
    total.backward()
    return total, depth, pose, trans, parts

l64, d64, p64, t64, parts64 = oracle(torch.float64)
l32, d32, p32, t32, _ = oracle(torch.float32)
g = synth.to_device(inp, dev)
dg = [d.requires_grad_(True) for d in g["depth"]]
tg = [t.requires_grad_(True) for t in g["trans"]]
pg = g["pose"].requires_grad_(True)
lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], dg, pg, tg, fused=True)
lg.backward(); torch.cuda.synchronize()
print("loss hip %.9f o32 %.9f o64 %.9f" % (float(lg), float(l32), float(l64)))
def st(a, b, name, excl=None):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    e = (a - b).abs(); sc = float(b.abs().max())
    msg = f"{name}: max {float(e.max())/sc:.2e} relL2 {float((a-b).norm()/b.norm()):.2e} frac>1e-4 {float((e>1e-4*sc).double().mean()):.2e}"
    if excl is not None:
        keep = ~excl
        msg += f" | excl {float(excl.double().mean()):.2e}: max {float(e[keep].max())/sc:.2e} relL2 {float((a-b)[keep].norm()/b[keep].norm()):.2e}"
    print(msg)
st(pg.grad, p64.grad, "pose hip-vs-64"); st(p32.grad, p64.grad, "pose o32-vs-64"); st(pg.grad, p32.grad, "pose hip-vs-32")
for s in range(S):
    tgt64 = inp["tgt"].double()
    es = []
    for i in range(2):
        G = parts64[s * N + i]["proj_image"].detach()
        es.append(0.15 * (G - tgt64).abs() + 0.85 * R.tf_ssim(G, tgt64, size=3, mean_metric=False))
    tie = ((es[0] - es[1]).abs() <= 1e-4 * torch.maximum(es[0], es[1])).any(dim=3)   # [B,H,W]
    tie = torch.nn.functional.max_pool2d(tie.float()[:, None], 3, 1, 1)[:, 0] > 0
    st(dg[s].grad.squeeze(3), d64[s].grad.squeeze(3), f"d_depth{s} hip-vs-64", tie)
    st(dg[s].grad.squeeze(3), d32[s].grad.squeeze(3), f"d_depth{s} hip-vs-32", tie)
    st(tg[s].grad, t64[s].grad, f"d_trans{s} hip-vs-64")
    st(tg[s].grad, t32[s].grad, f"d_trans{s} hip-vs-32")
