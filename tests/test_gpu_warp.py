"""GPU parity of the stand-alone operators against the CPU oracle (oracle/ref_graph.py),
through the C ABI.  Tolerances: outputs 1e-5 relative, gradients 1e-4 relative, projected
integer pixel indices bit-exact (BASELINE.json north_star).  Parity is unpinned by the
reference itself (no TF, no fixtures) - see the oracle's header."""
import pytest
import torch

import sfm_amd
from oracle import ref_graph as R
from helpers import CASES, TOL_GRAD, TOL_OUT, rel_err, warp_case

pytestmark = pytest.mark.gpu


def _dev(c, dev):
    return {k: (v.to(dev) if v is not None else None) for k, v in c.items()}


@pytest.mark.parametrize("B,H,W,ps", CASES)
@pytest.mark.parametrize("add_trans", [True, False])
def test_projected_coords_and_indices_bit_exact(dev, B, H, W, ps, add_trans):
    c = warp_case(B, H, W, seed=1, pose_scale=ps, add_trans=add_trans)
    g = _dev(c, dev)
    coords_o, z_o = R.project_coords(c["depth"], c["pose"], c["K"], False, add_trans, c["trans"])
    coords_g, z_g = sfm_amd.project_coords(g["depth"], g["pose"], g["K"], False, add_trans, g["trans"])
    _, ints = R.corner_indices(coords_o, H, W)
    idx_o = torch.cat(ints, dim=3)  # x0, x1, y0, y1
    idx_g = sfm_amd.corner_indices((B, H, W, 3), coords_g).cpu()
    assert torch.equal(idx_g, idx_o), f"{(idx_g != idx_o).sum().item()} corner indices differ"
    # the coordinates themselves follow the canonical fp32 op order: bitwise equal
    assert torch.equal(coords_g.cpu(), coords_o)
    assert torch.equal(z_g.cpu(), z_o)


def test_inverse_pose_indices_bit_exact(dev):
    c = warp_case(2, 16, 24, seed=3, pose_scale=0.05)
    g = _dev(c, dev)
    coords_o, _ = R.project_coords(c["depth"], c["pose"], c["K"], True, True, c["trans"])
    coords_g, _ = sfm_amd.project_coords(g["depth"], g["pose"], g["K"], True, True, g["trans"])
    assert torch.equal(coords_g.cpu(), coords_o)


@pytest.mark.parametrize("B,H,W,ps", CASES)
def test_projective_inverse_warp_forward_backward(dev, B, H, W, ps):
    c = warp_case(B, H, W, seed=2, pose_scale=ps)
    leaves_o = [c[k].clone().requires_grad_(True) for k in ("src", "depth", "pose", "trans")]
    out_o, mask_o = R.projective_inverse_warp(leaves_o[0], leaves_o[1], leaves_o[2], c["K"], add_trans=True,
                                               translations=leaves_o[3])
    wgt = torch.randn(out_o.shape, generator=torch.Generator().manual_seed(5))
    (out_o * wgt).sum().backward()
    leaves_g = [c[k].to(dev).requires_grad_(True) for k in ("src", "depth", "pose", "trans")]
    out_g, mask_g = sfm_amd.projective_inverse_warp(leaves_g[0], leaves_g[1], leaves_g[2], c["K"].to(dev),
                                                    add_trans=True, translations=leaves_g[3])
    (out_g * wgt.to(dev)).sum().backward()
    assert tuple(mask_g.shape) == (B, H, W, 1)
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert torch.equal(mask_g.cpu(), mask_o)
    for lg, lo, name in zip(leaves_g, leaves_o, ("img", "depth", "pose", "trans")):
        assert rel_err(lg.grad, lo.grad) <= TOL_GRAD, name


@pytest.mark.parametrize("B,H,W,ps", CASES)
def test_projective_warp_forward_backward(dev, B, H, W, ps):
    c = warp_case(B, H, W, seed=4, pose_scale=ps)
    leaves_o = [c[k].clone().requires_grad_(True) for k in ("tgt", "depth", "pose", "trans")]
    out_o, mask_o = R.projective_warp(leaves_o[0], leaves_o[1], leaves_o[2], c["K"], add_trans=True,
                                      translations=leaves_o[3])
    wgt = torch.randn(out_o.shape, generator=torch.Generator().manual_seed(6))
    (out_o * wgt).sum().backward()
    leaves_g = [c[k].to(dev).requires_grad_(True) for k in ("tgt", "depth", "pose", "trans")]
    out_g, mask_g = sfm_amd.projective_warp(leaves_g[0], leaves_g[1], leaves_g[2], c["K"].to(dev), add_trans=True,
                                            translations=leaves_g[3])
    (out_g * wgt.to(dev)).sum().backward()
    assert tuple(mask_g.shape) == (B, H, W, 3)
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert rel_err(mask_g, mask_o) <= TOL_OUT
    for lg, lo, name in zip(leaves_g, leaves_o, ("img", "depth", "pose", "trans")):
        assert rel_err(lg.grad, lo.grad) <= TOL_GRAD, name


@pytest.mark.parametrize("inverse_pose,add_trans", [(True, True), (True, False), (False, False)])
def test_warp_variants(dev, inverse_pose, add_trans):
    c = warp_case(2, 16, 24, seed=7, pose_scale=0.03, add_trans=add_trans)
    for fn_o, fn_g, img in ((R.projective_inverse_warp, sfm_amd.projective_inverse_warp, "src"),
                            (R.projective_warp, sfm_amd.projective_warp, "tgt")):
        d_o, p_o = c["depth"].clone().requires_grad_(True), c["pose"].clone().requires_grad_(True)
        out_o, _ = fn_o(c[img], d_o, p_o, c["K"], inverse_pose=inverse_pose, add_trans=add_trans,
                        translations=c["trans"])
        out_o.pow(2).sum().backward()
        d_g, p_g = c["depth"].to(dev).requires_grad_(True), c["pose"].to(dev).requires_grad_(True)
        out_g, _ = fn_g(c[img].to(dev), d_g, p_g, c["K"].to(dev), inverse_pose=inverse_pose, add_trans=add_trans,
                        translations=c["trans"].to(dev) if add_trans else None)
        out_g.pow(2).sum().backward()
        assert rel_err(out_g, out_o) <= TOL_OUT
        assert rel_err(d_g.grad, d_o.grad) <= TOL_GRAD
        assert rel_err(p_g.grad, p_o.grad) <= TOL_GRAD


@pytest.mark.parametrize("C", [1, 3, 5])
def test_bilinear_sampler_any_channels_and_sizes(dev, C):
    g = torch.Generator().manual_seed(11)
    B, Hs, Ws, Ht, Wt = 2, 9, 14, 7, 11
    img = torch.randn(B, Hs, Ws, C, generator=g)
    coords = torch.rand(B, Ht, Wt, 2, generator=g) * torch.tensor([Ws + 6.0, Hs + 6.0]) - 3.0  # incl. out of bounds
    coords[0, 0, 0] = torch.tensor([Ws - 1.0, Hs - 1.0])  # quirk 6: exactly on the last pixel
    coords[0, 0, 1] = torch.tensor([-1.0, 2.5])
    io, co = img.clone().requires_grad_(True), coords.clone().requires_grad_(True)
    out_o, mask_o = R.bilinear_sampler(io, co)
    out_o.pow(2).sum().backward()
    ig, cg = img.to(dev).requires_grad_(True), coords.to(dev).requires_grad_(True)
    out_g, mask_g = sfm_amd.bilinear_sampler(ig, cg)
    out_g.pow(2).sum().backward()
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert torch.equal(mask_g.cpu(), mask_o)
    assert rel_err(cg.grad, co.grad) <= TOL_GRAD
    assert rel_err(ig.grad, io.grad) <= TOL_GRAD


@pytest.mark.parametrize("C,use_depth", [(3, True), (3, False), (2, True)])
def test_bilinear_project_any_channels(dev, C, use_depth):
    g = torch.Generator().manual_seed(12)
    B, H, W = 2, 10, 13
    img = torch.randn(B, H, W, C, generator=g)
    base = torch.stack(torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                                      indexing="xy"), dim=-1)
    coords = base[None] + torch.randn(B, H, W, 2, generator=g) * 2.0
    z = torch.rand(B, H, W, 1, generator=g) * 60.0 - 5.0  # crosses both clip bounds of z/4 in [0,10]
    io, co, zo = img.clone().requires_grad_(True), coords.clone().requires_grad_(True), z.clone().requires_grad_(True)
    out_o, mask_o = R.bilinear_project(io, co, use_depth=use_depth, depth=zo)
    wgt = torch.randn(out_o.shape, generator=g)
    (out_o * wgt).sum().backward()
    ig, cg, zg = (t.to(dev).requires_grad_(True) for t in (img, coords, z))
    out_g, mask_g = sfm_amd.bilinear_project(ig, cg, use_depth=use_depth, depth=zg)
    (out_g * wgt.to(dev)).sum().backward()
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert rel_err(mask_g, mask_o) <= TOL_OUT
    assert rel_err(cg.grad, co.grad) <= TOL_GRAD
    assert rel_err(ig.grad, io.grad) <= TOL_GRAD
    if use_depth:
        assert rel_err(zg.grad, zo.grad) <= TOL_GRAD


@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 5, 7), (1, 128, 416)])
def test_tf_ssim(dev, B, H, W):
    g = torch.Generator().manual_seed(13)
    a = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    b = torch.clamp(a + 0.3 * torch.randn(B, H, W, 3, generator=g), -1, 1)
    for mean_metric in (False, True):
        ao, bo = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        lo = R.tf_ssim(ao, bo, mean_metric=mean_metric)
        lo.sum().backward()
        ag, bg = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        lg = sfm_amd.tf_ssim(ag, bg, mean_metric=mean_metric)
        lg.sum().backward()
        # the fp32 SSIM map is ill-conditioned where the local variance is ~C2 (m11 - m1^2
        # cancels); both sides are compared with the fp64 oracle at the same tolerance
        l64 = R.tf_ssim(a.double(), b.double(), mean_metric=mean_metric)
        tol = max(TOL_OUT, 2.0 * rel_err(lo, l64))
        assert rel_err(lg, l64) <= tol
        assert rel_err(ag.grad, ao.grad) <= max(TOL_GRAD, tol * 10)
        assert rel_err(bg.grad, bo.grad) <= max(TOL_GRAD, tol * 10)


def test_known_answers(dev):
    """Analytic pins: constant image through both warps."""
    c = warp_case(1, 16, 24, seed=21, pose_scale=0.01)
    g = _dev(c, dev)
    const = torch.full((1, 16, 24, 3), 0.37, device=dev)
    out, mask = sfm_amd.projective_warp(const, g["depth"], g["pose"], g["K"], add_trans=True, translations=g["trans"])
    hit = out != 0
    assert hit.any()
    assert (out[hit] - 0.37).abs().max() < 1e-6  # splat of a constant is that constant wherever anything landed
    out, mask = sfm_amd.projective_inverse_warp(const, g["depth"], g["pose"], g["K"], add_trans=True,
                                                translations=g["trans"])
    inside = out[:, 2:-2, 2:-2]
    assert (inside - 0.37).abs().max() < 1e-5  # interior taps: weights sum to 1
