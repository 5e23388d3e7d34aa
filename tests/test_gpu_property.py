"""Property tests (hypothesis, derandomised) of the stand-alone operators against the CPU oracle over random
intrinsics / poses / depths INCLUDING the degenerate ones the fixed cases of test_gpu_warp.py do not reach:
depth <= 0, points behind the camera, z / 4 beyond its clip, coordinates far outside the image or exactly on its
border, rotations near the +-pi clip (SURVEY.md 8c pin 4).  Bars as everywhere: coordinates and integer corner
indices bit-exact, outputs 1e-5 relative, masks equal."""
import math

import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import sfm_amd
from oracle import ref_graph as R
from helpers import TOL_OUT, rel_err

pytestmark = pytest.mark.gpu
SET = dict(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])


def _same(a, b):
    """bitwise equality with NaN == NaN"""
    a, b = a.cpu(), b.cpu()
    return bool(((a == b) | (a.isnan() & b.isnan())).all())


def _problem(seed, B, H, W, pose_scale, depth_lo, depth_hi, focal):
    g = torch.Generator().manual_seed(seed)
    depth = torch.rand(B, H, W, generator=g) * (depth_hi - depth_lo) + depth_lo
    depth[0, 0, 0] = 0.0                                  # a point AT the camera centre
    pose = torch.randn(B, 6, generator=g) * pose_scale
    pose[0, 3] = math.pi * (1.0 + pose_scale)             # beyond the clip of euler2mat (utils.py:37-39)
    K = torch.tensor([[focal, 0.0, W / 2.0 + 0.3], [0.0, focal * 1.02, H / 2.0 - 0.2], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
    trans = torch.randn(B, H, W, 3, generator=g) * 0.05
    img = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    return img, depth, pose, K, trans


@settings(**SET)
@given(seed=st.integers(0, 2 ** 20), B=st.integers(1, 3), H=st.integers(2, 18), W=st.integers(2, 22),
       pose_scale=st.sampled_from([1e-3, 0.05, 0.5, 2.0]), depth_lo=st.sampled_from([-3.0, 0.0, 0.1]),
       focal=st.sampled_from([3.0, 40.0, 700.0]), inverse=st.booleans(), add_trans=st.booleans())
def test_projected_coordinates_and_indices_bit_exact(dev, seed, B, H, W, pose_scale, depth_lo, focal, inverse, add_trans):
    _, depth, pose, K, trans = _problem(seed, B, H, W, pose_scale, depth_lo, 12.0, focal)
    co, zo = R.project_coords(depth, pose, K, inverse, add_trans, trans if add_trans else None)
    cg, zg = sfm_amd.project_coords(depth.to(dev), pose.to(dev), K.to(dev), inverse, add_trans,
                                    trans.to(dev) if add_trans else None)
    assert _same(cg, co) and _same(zg, zo)
    finite = torch.isfinite(co).all(dim=3, keepdim=True)
    _, ints = R.corner_indices(torch.where(finite, co, torch.zeros_like(co)), H, W)
    idx_g = sfm_amd.corner_indices((B, H, W, 3), torch.where(finite.to(dev), cg, torch.zeros_like(cg))).cpu()
    assert torch.equal(idx_g, torch.cat(ints, dim=3))


@settings(**SET)
@given(seed=st.integers(0, 2 ** 20), B=st.integers(1, 2), Hs=st.integers(1, 12), Ws=st.integers(1, 15), Ht=st.integers(1, 9),
       Wt=st.integers(1, 11), C=st.sampled_from([1, 3, 4]), spread=st.sampled_from([0.5, 4.0, 1e3, 1e6]))
def test_bilinear_sampler_forward_anywhere(dev, seed, B, Hs, Ws, Ht, Wt, C, spread):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, Hs, Ws, C, generator=g)
    coords = (torch.rand(B, Ht, Wt, 2, generator=g) - 0.5) * 2 * spread + torch.tensor([Ws / 2.0, Hs / 2.0])
    coords[0, 0, 0] = torch.tensor([Ws - 1.0, Hs - 1.0])      # exactly the last pixel (quirk 6)
    coords.view(-1, 2)[-1] = torch.tensor([-1.0, float(Hs)])  # exactly on the mask's border
    flat = coords.view(-1, 2)
    flat[: flat.shape[0] // 3] = flat[: flat.shape[0] // 3].round()   # integer coordinates: zero-weight taps
    out_o, mask_o = R.bilinear_sampler(img, coords)
    out_g, mask_g = sfm_amd.bilinear_sampler(img.to(dev), coords.to(dev))
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert torch.equal(mask_g.cpu(), mask_o)


@settings(**SET)
@given(seed=st.integers(0, 2 ** 20), B=st.integers(1, 2), H=st.integers(1, 12), W=st.integers(1, 15), C=st.sampled_from([1, 3]),
       jitter=st.sampled_from([0.3, 3.0, 40.0]), zscale=st.sampled_from([1.0, 30.0, 200.0]), use_depth=st.booleans())
def test_bilinear_project_forward_anywhere(dev, seed, B, H, W, C, jitter, zscale, use_depth):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, H, W, C, generator=g)
    base = torch.stack(torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                                      indexing="xy"), dim=-1)
    coords = base[None] + torch.randn(B, H, W, 2, generator=g) * jitter      # many land outside, many collide
    z = (torch.rand(B, H, W, 1, generator=g) - 0.3) * zscale                 # negative, and z / 4 > 10
    out_o, mask_o = R.bilinear_project(img, coords, use_depth=use_depth, depth=z)
    out_g, mask_g = sfm_amd.bilinear_project(img.to(dev), coords.to(dev), use_depth=use_depth, depth=z.to(dev))
    assert rel_err(out_g, out_o) <= TOL_OUT
    assert rel_err(mask_g, mask_o) <= TOL_OUT
    # the splat is linear in the image (its normaliser is not a function of it)
    img2 = torch.randn(B, H, W, C, generator=g)
    a = sfm_amd.bilinear_project((0.7 * img + 1.3 * img2).to(dev), coords.to(dev), use_depth=use_depth, depth=z.to(dev))[0]
    b = 0.7 * out_g + 1.3 * sfm_amd.bilinear_project(img2.to(dev), coords.to(dev), use_depth=use_depth, depth=z.to(dev))[0]
    assert rel_err(a, b, floor=1e-3) <= 2e-5


@settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 20), H=st.integers(4, 20), W=st.integers(4, 26), pose_scale=st.sampled_from([0.01, 0.3, 1.5]),
       depth_lo=st.sampled_from([-1.0, 0.05]), focal=st.sampled_from([8.0, 250.0]))
def test_both_warps_forward_degenerate_geometry(dev, seed, H, W, pose_scale, depth_lo, focal):
    img, depth, pose, K, trans = _problem(seed, 2, H, W, pose_scale, depth_lo, 8.0, focal)
    for fo, fg in ((R.projective_inverse_warp, sfm_amd.projective_inverse_warp), (R.projective_warp, sfm_amd.projective_warp)):
        out_o, mask_o = fo(img, depth, pose, K, add_trans=True, translations=trans)
        out_g, mask_g = fg(img.to(dev), depth.to(dev), pose.to(dev), K.to(dev), add_trans=True, translations=trans.to(dev))
        ok = torch.isfinite(out_o)
        assert torch.equal(torch.isfinite(out_g).cpu(), ok)
        assert rel_err(torch.where(ok.to(dev), out_g, torch.zeros_like(out_g)), torch.where(ok, out_o, torch.zeros_like(out_o))) <= TOL_OUT
        mk = torch.isfinite(mask_o)
        assert rel_err(torch.where(mk.to(dev), mask_g, torch.zeros_like(mask_g)), torch.where(mk, mask_o, torch.zeros_like(mask_o))) <= TOL_OUT
