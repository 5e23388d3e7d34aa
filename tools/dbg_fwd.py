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
NE=B*H*W*3
p = eng.forward_partials(tgt, src, K, depth, pose, trans, NE, True); torch.cuda.synchronize()
nf, mx = eng.overflow_state()
n_acc=S*N*B*H*W*4
ovf = eng.ws[n_acc:2*n_acc].view(-1,4)
print("flags raised", nf, "of", S*N*B, "ovf pixels nonzero frac", float((ovf[:,3]!=0).float().mean()))
shift_off = None
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
for _ in range(5): eng.forward_partials(tgt, src, K, depth, pose, trans, NE, True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): eng.forward_partials(tgt, src, K, depth, pose, trans, NE, True)
e1.record(); torch.cuda.synchronize()
print("SFM_DBG", os.environ.get("SFM_DBG"), "fwd us", e0.elapsed_time(e1)/50*1e3)
n_img = S*N*B; stx=(W+63)//64; sty=(H+15)//16
fl = (n_img+3)//4*4; sh = n_img*stx*sty*4
tab = eng.ws[-(fl+sh):-fl].view(torch.int32).view(-1,4).cpu()
import collections
print("shift x", tab[:,0].float().mean().item(), tab[:,0].min().item(), tab[:,0].max().item(), "y", tab[:,1].min().item(), tab[:,1].max().item())
print("reach x hist", sorted(collections.Counter(tab[:,2].tolist()).items()))
print("reach y hist", sorted(collections.Counter(tab[:,3].tolist()).items()))
scan = ((64+2*tab[:,2]).float()*(16+2*tab[:,3]).float()).mean().item()
print("mean scan pixels per tile", scan, "redundancy", scan/1024)
