"""Fused loss step at the per-GPU slices of BASELINE.json configs[1..3] (1 GPU): ms/step, Gpix/s, fraction of
the 8 TB/s roofline at SURVEY 8(d)'s 260 B per (pixel, scale, source)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_amd import synth
from sfm_amd.loss import LossEngine
dev = torch.device("cuda:0")
CASES = [("c2 416x128 B=8 S=4 N=2", 8, 128, 416, 4, 2, False),
         ("c3 1248x384 B=4/GPU (32 over 8 GPUs) S=4 N=2", 4, 384, 1248, 4, 2, False),
         ("c3 1248x384 B=16/GPU (32 over 2 GPUs) S=4 N=2", 16, 384, 1248, 4, 2, False),
         ("c4 2048x1024 B=8/GPU (64 over 8 GPUs) S=4 N=4", 8, 1024, 2048, 4, 4, True)]
if os.environ.get("SFM_CASES"):  # e.g. SFM_CASES=1,2
    CASES = [CASES[int(k)] for k in os.environ["SFM_CASES"].split(",")]
for label, B, H, W, S, N, city in CASES:
    inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, cityscapes=city), dev)
    tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
    depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
    trans = torch.stack(inp["trans"], 0).contiguous()
    del inp
    eng = LossEngine(B, H, W, S, N, dev)
    ne = B * H * W * 3
    def step():
        p = eng.forward_partials(tgt, src, K, depth, pose, trans, ne, True)
        return eng.backward(tgt, src, K, depth, pose, trans, p, ne, 1.0, True)
    for _ in range(3): out = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if B * H * W < 4e6 else 5
    for _ in range(n): out = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    units = B * H * W * S * N * 2
    eng.forward_partials(tgt, src, K, depth, pose, trans, ne, True); torch.cuda.synchronize()
    nf, _ = eng.overflow_state()  # flags raised by a forward (the backward lowers them)
    eng.backward(tgt, src, K, depth, pose, trans, eng.forward_partials(tgt, src, K, depth, pose, trans, ne, True), ne, 1.0, True)
    print(f"   overflow flags raised: {nf} of {S * N * B} images", flush=True)
    import ctypes
    from sfm_amd import _lib
    lib = _lib.load(); st = (ctypes.c_float * lib.sfm_prof_stages())()
    lib.sfm_prof_enable(1)
    for _ in range(3):
        step(); torch.cuda.synchronize(); lib.sfm_prof_read(st)
    lib.sfm_prof_enable(0)
    names = ["tile_shift", "warp_fwd", "stencil", "reduce", "seed_guard", "warp_bwd", "pose_grad", "mask_sums"]
    print("   stage us per Mpix-scale:", {n: round(v / 3 * 1e3 / (B * H * W * S / 1e6), 1) for n, v in zip(names, st)}, flush=True)
    print(f"{label}: {dt*1e3:.3f} ms/step  {units/dt/1e9:.2f} Gpix/s  step_frac_of_hbm_peak {130.0*units/dt/8e12:.3f}  "
          f"workspace {eng.ws.numel()*4/2**30:.2f} GiB  loss {float(out[0]):.5f}", flush=True)
    del eng, tgt, src, depth, trans, out
    from sfm_amd import loss as L; L._engines.clear(); torch.cuda.empty_cache()
