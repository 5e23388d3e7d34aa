"""One ASPP convolution in a loop (for rocprofv3 counter passes).  usage: python3 tools/one_conv.py h w Cin Cout rate [fwd|bwd_data|bwd_weight]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import _lib
from sfm_amd._lib import call, ptr, stream_of
h, w_, ci, co, r = (int(v) for v in sys.argv[1:6])
what = sys.argv[6] if len(sys.argv) > 6 else "fwd"
B = 8
dev = torch.device("cuda:0")
x = torch.randn(B, h, w_, ci, device=dev); wt = torch.randn(3, 3, ci, co, device=dev) / (3 * ci ** 0.5)
go = torch.randn(B, h, w_, co, device=dev)
y = torch.empty(B, h, w_, co, device=dev); gx = torch.empty_like(x); gw = torch.empty_like(wt)
ws = torch.empty(_lib.load().sfm_aspp_conv3x3_workspace_floats(B, h, w_, ci, co), device=dev)
st = stream_of(x)
import time
def loop(n):
  for _ in range(n):
      if what == "fwd": call("sfm_aspp_conv3x3_fwd", ptr(x), ptr(wt), B, h, w_, ci, co, r, ptr(y), ptr(ws), st)
      elif what == "bwd_data": call("sfm_aspp_conv3x3_bwd_data", ptr(go), ptr(wt), B, h, w_, ci, co, r, ptr(gx), ptr(ws), st)
      else: call("sfm_aspp_conv3x3_bwd_weight", ptr(x), ptr(go), B, h, w_, ci, co, r, ptr(gw), ptr(ws), st)
loop(5); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(50); torch.cuda.synchronize()
print(os.environ.get("TAG", ""), sys.argv[1:], "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
