"""Host-side cost of enqueueing one fused step (LossEngine.forward_partials + backward) - measured on a problem so
small that the GPU is never the limiter.  With world-size-1 nccl it also prices dist.all_reduce on the partials."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth
from sfm_amd.loss import LossEngine
import torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
B, H, W, S, N = 1, 16, 24, 4, 2
inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=1), dev)
tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous(); trans = torch.stack(inp["trans"], 0).contiguous()
eng = LossEngine(B, H, W, S, N, dev); ne = B * H * W * 3
def run(n, comm):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        p = eng.forward_partials(tgt, src, K, depth, pose, trans, ne, True)
        if comm == 1: dist.all_reduce(p)
        if comm == 2:
            h = dist.all_reduce(p, async_op=True); h.wait()
        out = eng.backward(tgt, src, K, depth, pose, trans, p, ne, 1.0, True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
run(50, 0); print(f"host us/step, no collective: {run(500, 0):.1f}")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
run(50, 1); print(f"host us/step, + dist.all_reduce (world 1, nccl): {run(500, 1):.1f}")
run(50, 2); print(f"host us/step, + async all_reduce + wait: {run(500, 2):.1f}")
dist.destroy_process_group()
