"""normalize_disp forward + backward alone at the c2 pyramid (per-kernel times under tools/prof_any.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfm_amd
dev = torch.device("cuda:0")
B, H, W, S = 8, 128, 416, 4
g = torch.Generator().manual_seed(1)
raw = [(10 * torch.sigmoid(torch.randn(B, H >> s, W >> s, 1, generator=g)) + 0.01).to(dev).requires_grad_(True) for s in range(S)]
gu = [torch.randn(r.shape, generator=g).to(dev) for r in raw]
gd = [torch.randn(r.shape, generator=g).to(dev) for r in raw]
only_u = len(sys.argv) > 1
for it in range(50):
    for r in raw:
        r.grad = None
    dn, dp = sfm_amd.normalize_disp(raw)
    if only_u:
        torch.autograd.backward(dn, gu)
    else:
        torch.autograd.backward(dn + dp, gu + gd)
torch.cuda.synchronize()
print("done")
