"""Would per-scale chains on separate streams overlap better than one S=4 launch sequence?  The scales of the loss are
independent until the final sum (kV depends on the sums of its own (scale, source) only), so S engines with S = 1 compute
the same gradients.  Both forms are captured as HIP graphs (the 4-chain form issues 4x the launches: eager it is bound by
the host) and replayed.  usage (GPU box): python3 tools/multistream_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth
from sfm_amd.loss import LossEngine
dev = torch.device("cuda:0")
B, H, W, S, N = 8, 128, 416, 4, 2
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED), dev)
tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
trans = torch.stack(inp["trans"], 0).contiguous()
ne = B * H * W * 3

def replay_time(g, n=200):
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

one = LossEngine(B, H, W, S, N, dev)
def step_one():
    p = one.forward_partials(tgt, src, K, depth, pose, trans, ne, True)
    return one.backward(tgt, src, K, depth, pose, trans, p, ne, 1.0, True)
for _ in range(3): step_one()
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    out1 = step_one()
print(f"one S=4 sequence, graph replay: {replay_time(g1):.4f} ms")

for groups in ([[0], [1], [2], [3]], [[0, 1], [2, 3]]):
    engs = [LossEngine(B, H, W, len(gs), N, dev) for gs in groups]
    dd = [depth[gs[0]:gs[-1] + 1].contiguous() for gs in groups]
    tt = [trans[gs[0]:gs[-1] + 1].contiguous() for gs in groups]
    streams = [torch.cuda.Stream() for _ in groups]
    def step_many():
        main = torch.cuda.current_stream()
        outs = []
        for e, d, t, st in zip(engs, dd, tt, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                p = e.forward_partials(tgt, src, K, d, pose, t, ne, True)
                outs.append(e.backward(tgt, src, K, d, pose, t, p, ne, 1.0, True))
        for st in streams:
            main.wait_stream(st)
        return outs
    for _ in range(3): step_many()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step_many()
    ms = replay_time(g)
    tot = sum(float(o[0]) for o in outs)
    print(f"{len(groups)} chains of {len(groups[0])} scale(s) on {len(groups)} streams, graph replay: {ms:.4f} ms   (sum of losses {tot:.5f} vs {float(out1[0]):.5f})")
