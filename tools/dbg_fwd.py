import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from sfm_amd import synth, _lib
from sfm_amd.loss import LossEngine
B,H,W,S,N=8,128,416,4,2
dev=torch.device('cuda:0')
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED), dev)
tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
trans = torch.stack(inp["trans"], 0).contiguous()
eng = LossEngine(B, H, W, S, N, dev)
p = eng.forward_partials(tgt, src, K, depth, pose, trans); torch.cuda.synchronize()
nf, mx = eng.overflow_state()
n_acc=S*N*B*H*W*4
ovf = eng.ws[n_acc:2*n_acc].view(-1,4)
print("flags raised", nf, "of", S*N*B, "ovf pixels nonzero frac", float((ovf[:,3]!=0).float().mean()))
shift_off = None
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
for _ in range(5): eng.forward_partials(tgt, src, K, depth, pose, trans)
torch.cuda.synchronize(); e0.record()
for _ in range(50): eng.forward_partials(tgt, src, K, depth, pose, trans)
e1.record(); torch.cuda.synchronize()
print("SFM_DBG", os.environ.get("SFM_DBG"), "fwd us", e0.elapsed_time(e1)/50*1e3)
