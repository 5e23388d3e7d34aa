import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth, _lib
from sfm_amd.loss import LossEngine
import ctypes
B,H,W,S,N=2,1024,2048,4,4
dev=torch.device('cuda:0')
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, cityscapes=True), dev)
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
n_img = S*N*B; stx=(W+63)//64; sty=(H+31)//32
fl = (n_img*stx*sty+3)//4*4; sh = n_img*stx*sty*4
tab = eng.ws[-(fl+sh):-fl].view(torch.int32).view(-1,4).cpu()
import collections
print("shift x", tab[:,0].min().item(), tab[:,0].max().item(), "y", tab[:,1].min().item(), tab[:,1].max().item())
print("reach x hist", sorted(collections.Counter(tab[:,2].tolist()).items()))
print("reach y hist", sorted(collections.Counter(tab[:,3].tolist()).items()))
print("redundancy", ((64+2*tab[:,2]).float()*(32+2*tab[:,3]).float()).mean().item()/2048)
lib=_lib.load(); st=(ctypes.c_float*lib.sfm_prof_stages())()
eng.backward(tgt, src, K, depth, pose, trans, p, NE, 1.0, True)
lib.sfm_prof_enable(1)
for _ in range(3):
    p = eng.forward_partials(tgt, src, K, depth, pose, trans, NE, True)
    eng.backward(tgt, src, K, depth, pose, trans, p, NE, 1.0, True); torch.cuda.synchronize(); lib.sfm_prof_read(st)
print("stage us", [round(v/3*1e3,1) for v in st], "order tile_shift warp_fwd stencil reduce finalize warp_bwd pose_grad")
