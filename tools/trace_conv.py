"""Cycle stamps of one workgroup's K-loop (library built with -DSFM_TRACE).  usage: python3 tools/trace_conv.py h w Cin Cout rate"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import _lib
from sfm_amd._lib import call, ptr, stream_of
h, w_, ci, co, r = (int(v) for v in sys.argv[1:6])
B = 8
dev = torch.device("cuda:0")
x = torch.randn(B, h, w_, ci, device=dev); wt = torch.randn(3, 3, ci, co, device=dev) / (3 * ci ** 0.5)
y = torch.empty(B, h, w_, co, device=dev)
lib = _lib.load()
ws = torch.empty(lib.sfm_aspp_conv3x3_workspace_floats(B, h, w_, ci, co), device=dev)
st = stream_of(x)
for _ in range(3): call("sfm_aspp_conv3x3_fwd", ptr(x), ptr(wt), B, h, w_, ci, co, r, ptr(y), ptr(ws), st)
torch.cuda.synchronize()
out = np.zeros(32768, dtype=np.int64)
lib.sfm_debug_trace.argtypes = [ctypes.c_void_p]
assert lib.sfm_debug_trace(out.ctypes.data) == 0
for blk, base in ((0, 0), (300, 8192)):
    for wave in range(8):
        t = out[base + wave * 1024: base + (wave + 1) * 1024]
        n = int((t != 0).sum()); t = t[:n]
        if n < 9: continue
        k = (n - 1) // 4
        q = t[:4 * k].reshape(k, 4)
        nxt = np.append(q[1:, 0], t[4 * k])
        issue = q[:, 1] - q[:, 0]; comp = q[:, 2] - q[:, 1]; wait = q[:, 3] - q[:, 2]; bar = nxt - q[:, 3]
        print(f"block {blk} {'multiplier' if wave < 4 else 'loader'} wave {wave}: {k} steps, total {t[-1]-t[0]} cyc; per step mean: issue {issue.mean():.0f} compute {comp.mean():.0f} "
              f"vmwait {wait.mean():.0f} barrier {bar.mean():.0f}  sum {(nxt - q[:,0]).mean():.0f}")
        if wave == 0 and blk == 0:
            print("   first 12 steps (issue, compute, vmwait, barrier):", [tuple(int(v) for v in row) for row in np.stack([issue, comp, wait, bar], 1)[:12]])

nb = int(os.environ.get("SFM_ASPP_NWG", "512"))
r = out[16384:16384 + 4 * nb].reshape(nb, 4)
t0 = r[:, 0].min()
st = (r[:, 0] - t0) / 100.0; en = (r[:, 1] - t0) / 100.0   # us (100 MHz)
cu = ((r[:, 3] & 15) << 16) | (r[:, 2] & 0xff00)            # xcc, se/sh/cu bits of HW_ID
print("kernel span %.1f us; starts: min %.1f max %.1f; durations: min %.1f mean %.1f max %.1f" % (en.max(), st.min(), st.max(), (en - st).min(), (en - st).mean(), (en - st).max()))
import collections
per = collections.Counter(cu.tolist())
print("distinct CUs", len(per), "blocks per CU histogram", collections.Counter(per.values()))
late = np.argsort(st)[-8:]
print("latest starters:", [(int(b), round(float(st[b]), 1), round(float(en[b]), 1)) for b in late])
print("start time percentiles (us):", [round(float(np.percentile(st, q)), 1) for q in (10, 25, 50, 75, 90, 99)])
print("end time percentiles (us):", [round(float(np.percentile(en, q)), 1) for q in (1, 10, 25, 50, 75, 90, 99)])
dur = en - st
ok = (dur > 0) & (dur < 1e4)
print("mean duration by XCD (blockIdx % 8):", [round(float(dur[ok & (np.arange(nb) % 8 == k)].mean()), 1) for k in range(8)])
print("mean duration by position in the XCD's range (8 bins):", [round(float(dur[ok & ((np.arange(nb) // 8) * 8 // (nb // 8) == k)].mean()), 1) for k in range(8)])
xcc = r[:, 3] & 15
print("XCC_ID of blocks 0..15:", xcc[:16].tolist())
