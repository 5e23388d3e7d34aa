"""GPU parity against the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py from the CPU oracle; the reference itself cannot run here, so
parity with it stays UNPINNED).  Nothing here needs the oracle or /root/reference at run time.
Tolerances (BASELINE.json): outputs/losses 1e-5 relative, gradients 1e-4, indices bit-exact."""
import os

import numpy as np
import pytest
import torch

import sfm_amd
from helpers import TOL_GRAD, TOL_OUT, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", ["warp_2x16x24", "warp_3x13x17_oob"])
def test_golden_warps(dev, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    K, wgt = _t(g["K"], dev), _t(g["wgt"], dev)
    coords, z = sfm_amd.project_coords(_t(g["depth"], dev), _t(g["pose"], dev), K, False, True, _t(g["trans"], dev))
    assert np.array_equal(coords.cpu().numpy(), g["coords"])
    assert np.array_equal(z.cpu().numpy(), g["z"])
    idx = sfm_amd.corner_indices(g["src"].shape, coords)
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    for kind, fn, img in (("gather", sfm_amd.projective_inverse_warp, "src"), ("splat", sfm_amd.projective_warp, "tgt")):
        leaves = [_t(g[k], dev).requires_grad_(True) for k in (img, "depth", "pose", "trans")]
        out, mask = fn(leaves[0], leaves[1], leaves[2], K, add_trans=True, translations=leaves[3])
        (out * wgt).sum().backward()
        assert rel_err(out, _t(g[f"{kind}_out"], dev)) <= TOL_OUT
        assert rel_err(mask, _t(g[f"{kind}_mask"], dev)) <= TOL_OUT
        for leaf, k in zip(leaves, ("img", "depth", "pose", "trans")):
            assert rel_err(leaf.grad, _t(g[f"{kind}_d_{k}"], dev)) <= TOL_GRAD, f"{kind} d_{k}"


@pytest.mark.parametrize("name", ["loss_2x16x24_s2n2", "loss_1x24x40_s3n4_pyr"])
@pytest.mark.parametrize("fused", [True, False])
def test_golden_loss(dev, name, fused):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, H, W, S, N = (int(v) for v in g["dims"])
    depth = [_t(g[f"depth{s}"], dev).requires_grad_(True) for s in range(S)]
    trans = [_t(g[f"trans{s}"], dev).requires_grad_(True) for s in range(S)]
    pose = _t(g["pose"], dev).requires_grad_(True)
    loss = sfm_amd.multi_scale_loss(_t(g["tgt"], dev), _t(g["src"], dev), _t(g["K"], dev), depth, pose, trans, fused=fused)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= TOL_OUT * abs(float(g["loss"]))
    assert rel_err(pose.grad, _t(g["d_pose"], dev)) <= TOL_GRAD
    for s in range(S):
        assert rel_err(depth[s].grad, _t(g[f"d_depth{s}"], dev)) <= TOL_GRAD, f"d_depth{s}"
        assert rel_err(trans[s].grad, _t(g[f"d_trans{s}"], dev)) <= TOL_GRAD, f"d_trans{s}"


def test_golden_partials_of_the_fused_forward(dev):
    """the S*(3N+1) sums that cross the data-parallel all-reduce"""
    from sfm_amd.loss import LossEngine
    g = np.load(os.path.join(GOLD, "loss_2x16x24_s2n2.npz"))
    B, H, W, S, N = (int(v) for v in g["dims"])
    depth = torch.stack([_t(g[f"depth{s}"], dev).squeeze(3) for s in range(S)], 0).contiguous()
    trans = torch.stack([_t(g[f"trans{s}"], dev) for s in range(S)], 0).contiguous()
    eng = LossEngine(B, H, W, S, N, dev)
    part = eng.forward_partials(_t(g["tgt"], dev), _t(g["src"], dev), _t(g["K"], dev), depth, _t(g["pose"], dev), trans,
                                B * H * W * 3, False)
    assert rel_err(part, _t(g["partials"], dev)) <= TOL_OUT


def test_golden_c1_bilinear_project_forward(dev):
    """BASELINE.json configs[0]: bilinear_project forward on one 416x128 3-frame snippet, B=1."""
    from sfm_amd import synth
    from helpers import warp_case
    g = np.load(os.path.join(GOLD, "c1_project_1x128x416.npz"))
    c = warp_case(1, 128, 416, seed=synth.SEED, pose_scale=0.01)
    sums = np.array([float(c[k].double().sum()) for k in ("tgt", "depth", "pose", "trans")])
    assert np.allclose(sums, g["input_sums"], rtol=1e-12)
    d = {k: v.to(dev) for k, v in c.items()}
    coords, z = sfm_amd.project_coords(d["depth"], d["pose"], d["K"], False, True, d["trans"])
    idx = sfm_amd.corner_indices((1, 128, 416, 3), coords)
    assert np.array_equal(idx.cpu().numpy().astype(np.int16), g["idx"])
    out, mask = sfm_amd.bilinear_project(d["tgt"], coords, use_depth=True, depth=z)
    assert rel_err(out, _t(g["out"], dev)) <= TOL_OUT and rel_err(mask, _t(g["mask"], dev)) <= TOL_OUT
    out2, mask2 = sfm_amd.projective_warp(d["tgt"], d["depth"], d["pose"], d["K"], add_trans=True, translations=d["trans"])
    assert rel_err(out2, _t(g["out"], dev)) <= TOL_OUT and rel_err(mask2, _t(g["mask"], dev)) <= TOL_OUT


def test_golden_pyramid_and_bn(dev):
    """SURVEY.md 8f rows 1-3 against committed numbers: normalise -> upsample -> regulariser, and batch norm + ELU."""
    g = np.load(os.path.join(GOLD, "pyramid_bn_2x16x24.npz"))
    B, H, W, S, N = (int(v) for v in g["dims"])
    raw = [_t(g[f"raw{s}"], dev).requires_grad_(True) for s in range(S)]
    exp = [_t(g[f"exp{s}"], dev).requires_grad_(True) for s in range(S)]
    dn, dp = sfm_amd.normalize_disp(raw)
    up_d = [sfm_amd.resize_bilinear(d, H, W) for d in dp]
    up_e = [sfm_amd.resize_bilinear(e, H, W) for e in exp]
    for s in range(S):
        assert rel_err(dn[s], _t(g[f"dispn{s}"], dev)) <= TOL_OUT and rel_err(dp[s], _t(g[f"depth{s}"], dev)) <= TOL_OUT
        assert rel_err(up_d[s], _t(g[f"up_depth{s}"], dev)) <= TOL_OUT
        assert np.array_equal(up_e[s].detach().cpu().numpy(), g[f"up_exp{s}"]), "same fp32 op order: bit-exact"
    sm = sum(0.01 / 2 ** s * sfm_amd.compute_smooth_loss(dn[s])
             + 0.01 / 2 ** s * (sfm_amd.compute_smooth_loss(exp[s]) + exp[s].abs().mean()) for s in range(S))
    assert rel_err(sm, _t(g["smooth"], dev)) <= TOL_OUT
    tot = sm + 1e-3 * sum((a * _t(g[f"wd{s}"], dev)).sum() for s, a in enumerate(up_d)) \
        + 1e-3 * sum((a * _t(g[f"we{s}"], dev)).sum() for s, a in enumerate(up_e))
    tot.backward()
    for s in range(S):
        assert rel_err(raw[s].grad, _t(g[f"d_raw{s}"], dev)) <= TOL_GRAD, f"d_raw{s}"
        assert rel_err(exp[s].grad, _t(g[f"d_exp{s}"], dev)) <= TOL_GRAD, f"d_exp{s}"
    x, gamma, beta = (_t(g[k], dev).requires_grad_(True) for k in ("bn_x", "bn_gamma", "bn_beta"))
    mm, mv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
    y = sfm_amd.batch_norm_elu(x, gamma, beta, mm, mv)
    y.backward(_t(g["bn_gy"], dev))
    assert rel_err(y, _t(g["bn_y"], dev)) <= TOL_OUT
    assert rel_err(mm, _t(g["bn_mm"], dev)) <= TOL_OUT and rel_err(mv, _t(g["bn_mv"], dev)) <= TOL_OUT
    for leaf, k in ((x, "bn_d_x"), (gamma, "bn_d_gamma"), (beta, "bn_d_beta")):
        assert rel_err(leaf.grad, _t(g[k], dev)) <= TOL_GRAD, k
