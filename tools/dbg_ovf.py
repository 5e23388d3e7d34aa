"""How local is the splat's overflow?  Per configuration: flagged images, fraction of pixels with a non-zero side
accumulator, and the fraction of 64x32 destination tiles that hold at least one such pixel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth
from sfm_amd.loss import LossEngine
dev = torch.device("cuda:0")
for label, B, H, W, S, N, city in [("c2", 8, 128, 416, 4, 2, False), ("c3", 4, 384, 1248, 4, 2, False), ("c4", 2, 1024, 2048, 4, 4, True)]:
    inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, cityscapes=city), dev)
    depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
    trans = torch.stack(inp["trans"], 0).contiguous()
    eng = LossEngine(B, H, W, S, N, dev)
    eng.forward_partials(inp["tgt"], inp["src"], inp["K"], depth, inp["pose"], trans, B * H * W * 3, True)
    torch.cuda.synchronize()
    n_acc = S * N * B * H * W * 4
    ovf = eng.ws[n_acc:2 * n_acc].view(S * N * B, H, W, 4)[..., 3] != 0
    th, tw = (H + 31) // 32, (W + 63) // 64
    pad = torch.zeros(S * N * B, th * 32, tw * 64, dtype=torch.bool, device=dev)
    pad[:, :H, :W] = ovf
    tiles = pad.view(-1, th, 32, tw, 64).any(dim=4).any(dim=2)
    print(label, "images with overflow", int(ovf.flatten(1).any(1).sum()), "of", S * N * B, " pixel frac %.2e" % float(ovf.float().mean()),
          " tile frac %.3f" % float(tiles.float().mean()), flush=True)
    del eng, inp, depth, trans
    torch.cuda.empty_cache()
