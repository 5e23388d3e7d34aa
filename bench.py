#!/usr/bin/env python3
"""Headline benchmark: warp + photometric loss, forward + backward, Gpix/s (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the fused hot path (sfm_loss_forward -> [all-reduce of the fp64
partial sums when N > 1] -> sfm_loss_backward) over one batch of synthetic KITTI-shaped
snippets that is already resident in HBM.  Workload (configs[1] of BASELINE.json):
416x128, B=8 per GPU, seq=3 (2 sources), 4 scales.  Weak scaling: every rank holds its
own 8 snippets; the loss is the global one.

Unit of work: a pixel-warp = one target pixel through one projective warp (gather or
splat direction) + its loss terms, forward and backward: B*H*W*S*N*2 per step and GPU.

The JSON line also carries
  roofline      the dominant kernel of the step: algorithmic bytes per launch / its mean
                launch duration (HIP events recorded by the library on the launch stream,
                in a second, instrumented pass of the same K steps) against 8 TB/s;
  cpu_baseline  the CPU oracle (oracle/ref_graph.py, "port" of the reference's TF graph
                to torch-CPU ops) timed on the host cores on the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, B, S, N = 128, 416, 8, 4, 2
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
STAGES = ["tile_shift", "warp_fwd", "stencil", "reduce", "finalize", "warp_bwd", "pose_grad"]
KERNEL_OF = {"warp_fwd": ["loss_splat_tile_kernel"], "stencil": ["loss_stencil_gather_kernel", "loss_stencil_splat_kernel"],
             "warp_bwd": ["loss_warp_bwd_kernel"]}


def algorithmic_bytes(n_src: int) -> dict:
    """Distinct fp32 tensor bytes each kernel must move per (pixel, scale) - DESIGN.md section 4."""
    n, g = n_src, min(n_src, 2)
    return {
        "warp_fwd": 4 + 12 + 12 * n + 12 * g + 12 * g + 16 * n,   # depth, tgt, trans, src taps; G out, Acc out
        "stencil": 16 * n + 12 * n + 12 * g + 12 + 32 * n + 12 * g,   # Acc, src, G, tgt in; Hb (2 parts), dG out
        "warp_bwd": 4 + 12 + 12 * n + 12 * g + 12 * g + 32 * n + 4 + 12 * n,  # + dG, Hb in; d_depth, d_trans out
    }


def cpu_baseline(seed: int) -> dict:
    from oracle import ref_graph as R
    from sfm_amd import synth
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=seed)

    def step():
        depth = [d.clone().requires_grad_(True) for d in inp["depth"]]
        trans = [t.clone().requires_grad_(True) for t in inp["trans"]]
        pose = inp["pose"].clone().requires_grad_(True)
        loss = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], depth, pose, trans)
        loss.backward()
        return float(loss)

    step()
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_all) < 25.0:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    units = B * H * W * S * N * 2
    return {"value": units / med / 1e9, "unit": "Gpix/s", "cores": cores, "kind": "port",
            "sample": f"full workload (B={B} {W}x{H} S={S} N={N}), fwd+bwd via torch autograd, median of {len(times)} "
                      f"steps after 1 warm-up ({med * 1e3:.0f} ms/step), reference-equivalent CPU graph "
                      f"(PyTorch-CPU restatement; TensorFlow is not installable here)"}


ASPP_LAYERS = [  # (h, w, Cin, Cout, rates) of disp_aspp_u_dense at 416x128 (nets.py:675-686)
    (8, 26, 1024, 512, (2, 4)), (16, 52, 512, 256, (2, 4)), (32, 104, 256, 128, (2, 4)),
    (64, 208, 128, 64, (3, 6)), (128, 416, 64, 32, (3, 6, 12))]
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak


def aspp_leg(dev, reps: int = 5) -> dict:
    """configs[4]: the 11 dilated 3x3 convolutions of the ASPP decoder at B=8, forward + backward data +
    backward weight through the C ABI, timed with events on the launch stream."""
    from sfm_amd import _lib
    from sfm_amd._lib import call, ptr, stream_of
    lib = _lib.load()
    jobs, flops = [], 0.0
    for h, w_, ci, co, rates in ASPP_LAYERS:
        x = torch.randn(B, h, w_, ci, device=dev)
        wt = torch.randn(3, 3, ci, co, device=dev) / (3 * ci ** 0.5)
        go = torch.randn(B, h, w_, co, device=dev)
        y, gx, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(wt)
        ws = torch.empty(lib.sfm_aspp_conv3x3_workspace_floats(B, h, w_, ci, co), device=dev)
        for r in rates:
            jobs.append((x, wt, go, y, gx, gw, ws, h, w_, ci, co, r))
            flops += 3 * 2.0 * B * h * w_ * co * 9 * ci

    def run():
        for x, wt, go, y, gx, gw, ws, h, w_, ci, co, r in jobs:
            st = stream_of(x)
            call("sfm_aspp_conv3x3_fwd", ptr(x), ptr(wt), B, h, w_, ci, co, r, ptr(y), ptr(ws), st)
            call("sfm_aspp_conv3x3_bwd_data", ptr(go), ptr(wt), B, h, w_, ci, co, r, ptr(gx), ptr(ws), st)
            call("sfm_aspp_conv3x3_bwd_weight", ptr(x), ptr(go), B, h, w_, ci, co, r, ptr(gw), ptr(ws), st)

    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = flops / (ms * 1e-3) / 1e12
    return {"workload": "11 dilated 3x3 convs of disp_aspp_u_dense, B=8 416x128, fwd + bwd-data + bwd-weight (configs[4])",
            "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
            "ms": ms, "flops": flops, "dtype": "f32"}


def total_loss_leg(dev, reps: int = 50) -> dict:
    """SURVEY.md 8(f) rows 1-2 widened path: total_loss of SfMLearner_uni.py:40-176 from the raw network outputs
    (disparity / translation pyramids at H/2^s, poses) through the Python mirror - disp normalisation, pyramid
    upsample, the fused step above, smoothness regulariser and every adjoint - host overhead included."""
    import sfm_amd
    from sfm_amd import synth
    inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED, full_res=False), dev)
    raw = [(d * 3.0).requires_grad_(True) for d in inp["disp"]]
    trans = [t.requires_grad_(True) for t in inp["trans"]]
    pose = inp["pose"].requires_grad_(True)

    def step():
        for t in raw + trans + [pose]:
            t.grad = None
        dn, dp = sfm_amd.normalize_disp(raw)
        loss = sfm_amd.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], dp, pose, trans, pred_disp=dn, smooth_weight=0.01)
        loss.backward()
        return loss

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # the same call sequence captured once as a HIP graph and replayed (inputs copied into its static buffers per call)
    g = sfm_amd.GraphedTotalLoss(inp["tgt"], inp["src"], inp["K"], raw, pose, trans, smooth_weight=0.01)
    raw_d, trans_d, pose_d = [t.detach() for t in raw], [t.detach() for t in trans], pose.detach()
    for _ in range(10):
        gl = g(raw_d, pose_d, trans_d)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gl = g(raw_d, pose_d, trans_d)[0]
    torch.cuda.synchronize()
    ms_graph = (time.perf_counter() - t0) / reps * 1e3
    graph_part = {"ms_per_call_hip_graph": ms_graph, "Gpix_per_s_hip_graph": B * H * W * S * N * 2 / (ms_graph * 1e-3) / 1e9,
                  "loss_hip_graph": float(gl.detach())}
    return {**graph_part, "workload": f"total_loss fwd+bwd from raw pyramids (scale s at {H}>>s x {W}>>s), B={B}, smooth_weight=0.01, "
                        f"through sfm_amd.multi_scale_loss (autograd, host overhead included)",
            "ms_per_call": ms, "Gpix_per_s": B * H * W * S * N * 2 / (ms * 1e-3) / 1e9, "loss": float(loss.detach())}


def config2_leg(dev, reps: int = 10) -> dict:
    """The per-GPU slice of BASELINE.json configs[2] at 8 GPUs (1248x384, B=32 -> 4 snippets per GPU): same fused step."""
    from sfm_amd import synth
    from sfm_amd.loss import LossEngine
    b2, h2, w2 = 4, 384, 1248
    inp = synth.to_device(synth.make_inputs(b2, h2, w2, S=S, N=N, seed=synth.SEED), dev)
    tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
    depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
    trans = torch.stack(inp["trans"], 0).contiguous()
    eng = LossEngine(b2, h2, w2, S, N, dev)
    ne = b2 * h2 * w2 * 3

    def step():
        p = eng.forward_partials(tgt, src, K, depth, pose, trans, ne, True)
        return eng.backward(tgt, src, K, depth, pose, trans, p, ne, 1.0, True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    units = b2 * h2 * w2 * S * N * 2
    res = {"workload": f"{w2}x{h2} B={b2}/GPU (configs[2]: B=32 over 8 GPUs) S={S} N={N}, same fused step, 1 GPU",
           "ms_per_step": dt * 1e3, "Gpix_per_s": units / dt / 1e9,
           "step_frac_of_hbm_peak": 130.0 * units / dt / 1e9 / HBM_PEAK_GBS, "loss": float(out[0])}
    del eng
    return res


def loader_leg(dev, reps: int = 50) -> dict:
    """decoded uint8 snippets resident in HBM -> tgt / src stack in [-1, 1] + multi-scale
    intrinsics, with the training augmentation (sfm_snippet_unpack: two launches)."""
    from sfm_amd.data_loader import AugmentDraws, snippet_unpack
    g = torch.Generator().manual_seed(8964)
    seq = torch.randint(0, 256, (B, H, (N + 1) * W, 3), dtype=torch.uint8, generator=g).to(dev)
    k = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    d = AugmentDraws(1.1, (0.9, 1.0, 1.1), True, 1.2)
    for _ in range(5):
        snippet_unpack(seq, k, H, W, N, S, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        snippet_unpack(seq, k, H, W, N, S, d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = B * H * W * (N + 1) * (3 * 2 + 12)  # uint8 read twice (sums, then apply), fp32 written once
    return {"workload": f"snippet unpack + augmentation + preprocess, B={B} {W}x{H} seq={N + 1}, through "
                        f"sfm_amd.snippet_unpack (host overhead included)",
            "ms_per_batch": ms, "snippets_per_s": B / (ms * 1e-3), "algorithmic_GBps": nbytes / (ms * 1e-3) / 1e9}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aspp", action="store_true")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="N > 1: two all-reduces hidden under kernels (LossEngine.step_dp) instead of the one "
                         "stream-ordered all-reduce between the two halves; costs ~46 us of extra launches and "
                         "stream hand-offs per step (measured with a 1-rank group), so it only pays when the "
                         "collective itself is slower than that")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # rehearsal aid: SFM_BENCH_FORCE_DP=1 runs the N > 1 sequence (process group, hidden all-reduces) on one GPU
    force_dp = world == 1 and os.environ.get("SFM_BENCH_FORCE_DP") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # RCCL prints a version banner on fd 1 when its first communicator comes up; stdout carries ONE JSON line,
        # so fd 1 points at stderr until the group and its first collective are through
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if force_dp:
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(8, dtype=torch.float64, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from sfm_amd import _lib, synth
    from sfm_amd.loss import LossEngine
    lib = _lib.load()

    inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=synth.SEED + rank), dev)
    tgt, src, K, pose = inp["tgt"], inp["src"], inp["K"], inp["pose"]
    depth = torch.stack([d.squeeze(3) for d in inp["depth"]], 0).contiguous()
    trans = torch.stack(inp["trans"], 0).contiguous()
    eng = LossEngine(B, H, W, S, N, dev)
    n_elem_global = world * B * H * W * 3

    overlapped = (world > 1 or force_dp) and args.overlap_allreduce

    def step_plain():
        partials = eng.forward_partials(tgt, src, K, depth, pose, trans, n_elem_global, True)
        if world > 1 or force_dp:
            dist.all_reduce(partials)
        return eng.backward(tgt, src, K, depth, pose, trans, partials, n_elem_global, 1.0, True)

    def step():
        if overlapped:  # both all-reduces hidden under kernels (LossEngine.step_dp)
            return eng.step_dp(tgt, src, K, depth, pose, trans, n_elem_global)[:4]
        return step_plain()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out[0])

    # second, instrumented pass: HIP events around every launch, on the launch stream
    stage_ms = (ctypes.c_float * len(STAGES))()
    lib.sfm_prof_enable(1)
    for _ in range(args.steps):
        step_plain()  # the per-stage events live in the plain sequence (same kernels, same launches)
        torch.cuda.synchronize()
        lib.sfm_prof_read(stage_ms)
    lib.sfm_prof_enable(0)

    if rank == 0:
        units_per_gpu = B * H * W * S * N * 2
        ms_step = dt / args.steps * 1e3
        value = units_per_gpu * world / (dt / args.steps) / 1e9
        per_stage_us = {n: stage_ms[i] / args.steps * 1e3 for i, n in enumerate(STAGES)}
        ab = algorithmic_bytes(N)
        dom = max(ab, key=lambda n: per_stage_us[n])
        launch_bytes = ab[dom] * B * H * W * S
        achieved = launch_bytes / (per_stage_us[dom] * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the TCC counters (tools/traffic.sh), same workload
            tj = json.load(open(tpath))
            if all(k in tj for k in KERNEL_OF[dom]):
                traffic = sum(tj[k]["bytes"] for k in KERNEL_OF[dom])
        step_bytes = (108 + 152) * B * H * W * S * N  # gather 108 B + splat 152 B per pair
        line = {
            "metric": "warp+loss fwd+bwd Gpix/s at 416x128 B=8 (1/2/4/8 GPU), % HBM roofline",
            "value": value, "unit": "Gpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{W}x{H} B={B}/GPU seq=3 (N={N} sources) S={S} scales, warp+photometric loss "
                                   f"fwd+bwd (BASELINE.json configs[1])",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "collectives": ("none" if world == 1 else ("2 all-reduces hidden under kernels" if overlapped
                                                                   else "1 stream-ordered all-reduce of S*(3N+1) fp64 sums")),
                       "pixel_warps_per_step": units_per_gpu * world,
                       "loss": loss},
            "roofline": {"bound": "hbm", "kernel": "+".join(KERNEL_OF[dom]), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": launch_bytes, "avg_launch_us": per_stage_us[dom],
                         "stage_us": per_stage_us,
                         "step_frac_of_hbm_peak": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_aspp:
            line["aspp"] = aspp_leg(dev)
            line["config"]["config5_step_ms"] = ms_step + line["aspp"]["ms"]  # ASPP convs + warp loss, fwd + bwd
        if world == 1 and not args.no_aspp:
            line["end_to_end"] = total_loss_leg(dev)
            line["loader"] = loader_leg(dev)
            line["config2_slice"] = config2_leg(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth.SEED)
        print(json.dumps(line), flush=True)
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
