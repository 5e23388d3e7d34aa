"""One decoder level (disp_bottleneck_dense) forward + backward at a config-5 size, for rocprofv3 --kernel-trace --stats.
usage: python3 tools/bench_level.py [cin cout h w]   (default 128 64 32 104: the level that ends at 64x208)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd
cin, cout, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 64, 32, 104)
B, dev = 8, torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
def conv(k, ci, co):
    return dict(w=(torch.randn(k, k, ci, co, generator=g) / (k * ci ** 0.5)).to(dev).requires_grad_(True),
                gamma=(torch.rand(co, generator=g) + 0.5).to(dev).requires_grad_(True),
                beta=(torch.randn(co, generator=g) * 0.1).to(dev).requires_grad_(True))
p = dict(up=conv(1, cin, cin), sup=conv(1, cin, cout), aspp_up=conv(3, cin, cout), aspp_up2=conv(3, 2 * cout, cout),
         out=conv(1, 5 * cout, cout), skip=conv(1, 96, cout), skip2=conv(1, cin, cout),
         skip_bn=dict(gamma=(torch.rand(96, generator=g) + 0.5).to(dev).requires_grad_(True),
                      beta=(torch.randn(96, generator=g) * 0.1).to(dev).requires_grad_(True)))
concat = torch.randn(B, h, w, cin, generator=g).to(dev).requires_grad_(True)
skip_feat = torch.randn(B, 2 * h, 2 * w, 96, generator=g).to(dev)
skip2 = torch.randn(B, h, w, cin, generator=g).to(dev)
wy = torch.randn(B, 2 * h, 2 * w, cout, generator=g).to(dev)
def step():
    y, u = sfm_amd.disp_bottleneck_dense(p, concat, skip_feat, (3, 6), True, True, skip2)
    (y * wy).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize()
print(f"level cin={cin} cout={cout} {h}x{w} -> {2*h}x{2*w}, B={B}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms fwd+bwd")
