#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference cannot be executed in the build image (it is TensorFlow-1.x graph code,
TensorFlow is not installed, `ModuleNotFoundError`) and ships no fixtures of its own,
so these vectors are produced by the CPU restatement (oracle/ref_graph.py, fp32 canonical
op order) - PARITY UNPINNED against the reference itself.  What they pin is (a) the
oracle against silent drift and (b) the HIP path against fixed, committed numbers that
travel to the GPU box (where neither /root/reference nor a need to re-run the oracle
exists).  Inputs come from sfm_amd.synth with the seeds below and are stored too.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_graph as R  # noqa: E402
from sfm_amd import synth  # noqa: E402
from helpers import warp_case  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def warp_fixture(B, H, W, seed, pose_scale):
    c = warp_case(B, H, W, seed=seed, pose_scale=pose_scale)
    out = {k: npy(v) for k, v in c.items()}
    coords, z = R.project_coords(c["depth"], c["pose"], c["K"], False, True, c["trans"])
    _, ints = R.corner_indices(coords, H, W)
    out.update(coords=npy(coords), z=npy(z), idx=npy(torch.cat(ints, dim=3)))
    wg = torch.randn(B, H, W, 3, generator=torch.Generator().manual_seed(seed + 100))
    out["wgt"] = npy(wg)
    for name, fn, img in (("gather", R.projective_inverse_warp, "src"), ("splat", R.projective_warp, "tgt")):
        leaves = [c[k].clone().requires_grad_(True) for k in (img, "depth", "pose", "trans")]
        o, m = fn(leaves[0], leaves[1], leaves[2], c["K"], add_trans=True, translations=leaves[3])
        (o * wg).sum().backward()
        out[f"{name}_out"], out[f"{name}_mask"] = npy(o), npy(m)
        for leaf, k in zip(leaves, ("img", "depth", "pose", "trans")):
            out[f"{name}_d_{k}"] = npy(leaf.grad)
    return out


def loss_fixture(B, H, W, S, N, seed, pose_scale, full_res):
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=seed, pose_scale=pose_scale, full_res=full_res)
    depth = [d.clone().requires_grad_(True) for d in inp["depth"]]
    trans = [t.clone().requires_grad_(True) for t in inp["trans"]]
    pose = inp["pose"].clone().requires_grad_(True)
    loss = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], depth, pose, trans)
    loss.backward()
    out = dict(tgt=npy(inp["tgt"]), src=npy(inp["src"]), K=npy(inp["K"]), pose=npy(inp["pose"]), loss=npy(loss),
               d_pose=npy(pose.grad), dims=np.array([B, H, W, S, N]))
    part = R.loss_partials(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    out["partials"] = npy(part)
    for s in range(S):
        out[f"depth{s}"], out[f"trans{s}"] = npy(inp["depth"][s]), npy(inp["trans"][s])
        out[f"d_depth{s}"], out[f"d_trans{s}"] = npy(depth[s].grad), npy(trans[s].grad)
    return out


def c1_fixture():
    """BASELINE.json configs[0]: bilinear_project forward (via projective_warp) on one 416x128 snippet, B=1.
    Inputs are regenerated from the seed (their checksums are stored); outputs are stored in full."""
    c = warp_case(1, 128, 416, seed=synth.SEED, pose_scale=0.01)
    coords, z = R.project_coords(c["depth"], c["pose"], c["K"], False, True, c["trans"])
    out, mask = R.bilinear_project(c["tgt"], coords, use_depth=True, depth=z)
    _, ints = R.corner_indices(coords, 128, 416)
    sums = np.array([float(c[k].double().sum()) for k in ("tgt", "depth", "pose", "trans")])
    return dict(out=npy(out), mask=npy(mask), idx=npy(torch.cat(ints, dim=3)).astype(np.int16), input_sums=sums)


def pyramid_fixture():
    """raw disparity pyramid -> normalise -> depth; bilinear upsample; smoothness regulariser;
    batch norm + ELU.  Inputs, outputs and gradients (against the stored upstream weights)."""
    g = torch.Generator().manual_seed(105)
    B, H, W, S, N = 2, 16, 24, 3, 2
    out = dict(dims=np.array([B, H, W, S, N]))
    raw = [(10.0 * torch.sigmoid(torch.randn(B, H >> s, W >> s, 1, generator=g)) + 0.01).requires_grad_(True) for s in range(S)]
    exp = [(0.01 * torch.randn(B, H >> s, W >> s, 3 * N, generator=g)).requires_grad_(True) for s in range(S)]
    dn, dp = R.normalize_disp(raw)
    up_d = [R.resize_bilinear_tf1(d, H, W) for d in dp]
    up_e = [R.resize_bilinear_tf1(e, H, W) for e in exp]
    sm = sum(0.01 / 2 ** s * R.compute_smooth_loss(dn[s]) + 0.01 / 2 ** s * (R.compute_smooth_loss(exp[s]) + exp[s].abs().mean())
             for s in range(S))
    wd = [torch.randn(B, H, W, 1, generator=g) for _ in range(S)]
    we = [torch.randn(B, H, W, 3 * N, generator=g) for _ in range(S)]
    (sm + sum((a * b).sum() for a, b in zip(up_d, wd)) * 1e-3 + sum((a * b).sum() for a, b in zip(up_e, we)) * 1e-3).backward()
    out["smooth"] = npy(sm)
    for s in range(S):
        out[f"raw{s}"], out[f"exp{s}"] = npy(raw[s]), npy(exp[s])
        out[f"dispn{s}"], out[f"depth{s}"] = npy(dn[s]), npy(dp[s])
        out[f"up_depth{s}"], out[f"up_exp{s}"] = npy(up_d[s]), npy(up_e[s])
        out[f"wd{s}"], out[f"we{s}"] = npy(wd[s]), npy(we[s])
        out[f"d_raw{s}"], out[f"d_exp{s}"] = npy(raw[s].grad), npy(exp[s].grad)
    x = (torch.randn(2, 6, 10, 32, generator=g) * 1.5 + 0.7).requires_grad_(True)
    gamma = (torch.rand(32, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(32, generator=g) * 0.3).requires_grad_(True)
    gy = torch.randn(2, 6, 10, 32, generator=g)
    mm, mv = torch.zeros(32), torch.ones(32)
    y = R.batch_norm_elu(x, gamma, beta, mm, mv)
    y.backward(gy)
    out.update(bn_x=npy(x), bn_gamma=npy(gamma), bn_beta=npy(beta), bn_gy=npy(gy), bn_y=npy(y), bn_mm=npy(mm), bn_mv=npy(mv),
               bn_d_x=npy(x.grad), bn_d_gamma=npy(gamma.grad), bn_d_beta=npy(beta.grad))
    return out


def loader_fixture():
    """decoded uint8 snippets [B, H, (N+1) W, 3] + intrinsics -> what oracle/ref_loader.py makes of
    them (validation path and one augmented path with fixed draws), for N = 2 and N = 4."""
    from oracle import ref_loader as L
    rng = np.random.default_rng(105)
    out = {}
    for tag, (B, H, W, N) in {"n2": (2, 12, 20, 2), "n4": (1, 9, 11, 4)}.items():
        low = rng.uniform(0, 255, (B, H // 3 + 2, (N + 1) * W // 3 + 2, 3))
        seq = np.clip(np.kron(low, np.ones((1, 3, 3, 1)))[:, :H, :(N + 1) * W] + rng.uniform(-15, 15, (B, H, (N + 1) * W, 3)),
                      0, 255).astype(np.uint8)
        k = np.tile(np.float32([[241.7, 0, 204.2], [0, 246.3, 59.0], [0, 0, 1]]), (B, 1, 1))
        draws = np.float32([1.17, 0.83, 1.1, 1.19, 1.0, 1.43])  # brightness, colours, swap, contrast
        vt, vs, km = L.fetch_batch(seq, k, H, W, N, 4, augment=False)
        at, as_, _ = L.fetch_batch(seq, k, H, W, N, 4, True, float(draws[0]), [float(c) for c in draws[1:4]], bool(draws[4]),
                                   float(draws[5]))
        out.update({f"{tag}_dims": np.array([B, H, W, N]), f"{tag}_seq": seq, f"{tag}_K": k, f"{tag}_draws": draws,
                    f"{tag}_val_tgt": vt, f"{tag}_val_src": vs, f"{tag}_K_ms": km, f"{tag}_aug_tgt": at, f"{tag}_aug_src": as_})
    return out


def main():
    torch.set_num_threads(4)
    np.savez_compressed(os.path.join(HERE, "loader_snippets.npz"), **loader_fixture())
    np.savez_compressed(os.path.join(HERE, "pyramid_bn_2x16x24.npz"), **pyramid_fixture())
    np.savez_compressed(os.path.join(HERE, "warp_2x16x24.npz"), **warp_fixture(2, 16, 24, 101, 0.05))
    np.savez_compressed(os.path.join(HERE, "warp_3x13x17_oob.npz"), **warp_fixture(3, 13, 17, 102, 0.2))
    np.savez_compressed(os.path.join(HERE, "loss_2x16x24_s2n2.npz"), **loss_fixture(2, 16, 24, 2, 2, 103, 0.02, True))
    np.savez_compressed(os.path.join(HERE, "loss_1x24x40_s3n4_pyr.npz"), **loss_fixture(1, 24, 40, 3, 4, 104, 0.02, False))
    np.savez_compressed(os.path.join(HERE, "c1_project_1x128x416.npz"), **c1_fixture())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
