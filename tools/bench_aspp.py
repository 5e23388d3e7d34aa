"""Per-layer timing of the 11 dilated 3x3 convolutions of disp_aspp_u_dense at config 5 (B=8, 416x128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd
from sfm_amd import _lib
from sfm_amd._lib import call, ptr, stream_of
LAYERS = [  # (h, w, Cin, Cout, rates)  nets.py:675-686 via SURVEY.md 8a row 14
    (8, 26, 1024, 512, (2, 4)), (16, 52, 512, 256, (2, 4)), (32, 104, 256, 128, (2, 4)),
    (64, 208, 128, 64, (3, 6)), (128, 416, 64, 32, (3, 6, 12))]
B = 8
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}; flops_total = 0.0
for h, w_, ci, co, rates in LAYERS:
    x = torch.randn(B, h, w_, ci, device=dev); wt = torch.randn(3, 3, ci, co, device=dev) / (3 * ci ** 0.5)
    go = torch.randn(B, h, w_, co, device=dev)
    y = torch.empty(B, h, w_, co, device=dev); gx = torch.empty_like(x); gw = torch.empty_like(wt)
    ws = torch.empty(_lib.load().sfm_aspp_conv3x3_workspace_floats(B, h, w_, ci, co), device=dev)
    fl = 2.0 * B * h * w_ * co * 9 * ci
    for r in rates:
        st = stream_of(x)
        t = {"fwd": timeit(lambda: call("sfm_aspp_conv3x3_fwd", ptr(x), ptr(wt), B, h, w_, ci, co, r, ptr(y), ptr(ws), st)),
             "bwd_data": timeit(lambda: call("sfm_aspp_conv3x3_bwd_data", ptr(go), ptr(wt), B, h, w_, ci, co, r, ptr(gx), ptr(ws), st)),
             "bwd_weight": timeit(lambda: call("sfm_aspp_conv3x3_bwd_weight", ptr(x), ptr(go), B, h, w_, ci, co, r, ptr(gw), ptr(ws), st))}
        for k in t: tot[k] += t[k]
        flops_total += fl
        print(f"{h}x{w_} {ci}->{co} r{r}: " + "  ".join(f"{k} {v*1e6:7.1f} us {fl/v/1e12:6.1f} TF" for k, v in t.items()), flush=True)
print("total: " + "  ".join(f"{k} {v*1e3:.3f} ms {flops_total/v/1e12:.1f} TF" for k, v in tot.items()),
      f" all {sum(tot.values())*1e3:.3f} ms {3*flops_total/sum(tot.values())/1e12:.1f} TF (peak 157.3)")

# batch norm + ELU after each convolution (csrc/bn_elu.hip): HBM-bound; algorithmic bytes = x in + y out (forward),
# x, y, gy in + gx out (backward); the kernels read x twice (statistics, then apply) and x, y, gy twice in the backward
print("batch norm + ELU (slim arg-scope epilogue), per conv output:")
tot_f = tot_b = 0.0
for h, w_, ci, co, rates in LAYERS:
    x = torch.randn(B, h, w_, co, device=dev, requires_grad=True)
    gamma = torch.ones(co, device=dev, requires_grad=True); beta = torch.zeros(co, device=dev, requires_grad=True)
    gy = torch.randn(B, h, w_, co, device=dev)
    y = sfm_amd.batch_norm_elu(x, gamma, beta)
    tf_ = timeit(lambda: sfm_amd.batch_norm_elu(x, gamma, beta))
    def fb():
        x.grad = None
        sfm_amd.batch_norm_elu(x, gamma, beta).backward(gy)
    tb_ = timeit(fb) - tf_
    nbytes = x.numel() * 4
    tot_f += tf_ * len(rates); tot_b += tb_ * len(rates)
    print(f"{h}x{w_} C={co}: fwd {tf_*1e6:7.1f} us {2*nbytes/tf_/1e9:7.0f} GB/s algorithmic   bwd {tb_*1e6:7.1f} us {4*nbytes/tb_/1e9:7.0f} GB/s algorithmic"
          f"   ({nbytes/1e6:.1f} MB per tensor; through the Python mirror, allocations included)")
print(f"all 11 epilogues: fwd {tot_f*1e3:.3f} ms  bwd {tot_b*1e3:.3f} ms")
