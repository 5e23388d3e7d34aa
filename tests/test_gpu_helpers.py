"""The geometry helpers utils.py exports (euler2mat, pose_vec2mat, meshgrid, pixel2cam, cam2pixel: utils.py:27-170) as
stand-alone HIP operators: forward BIT-EXACT against the oracle's canonical fp32 op order (these feed integer pixel
indices), gradients (hand-written adjoints, csrc/geom.hip) against autograd of the oracle at 1e-4, and the
reference's own composition utils.py:185-202 rebuilt from the pieces against the fused sfm_project_coords."""
import math

import pytest
import torch

import sfm_amd
from oracle import ref_graph as R
from sfm_amd import synth
from helpers import TOL_GRAD, rel_err

pytestmark = pytest.mark.gpu


def _angles(n, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, 1, generator=g) * 0.8
    a[0, 0] = math.pi * 1.25      # beyond the clip (utils.py:37-39)
    a[1, 0] = -math.pi * 1.5
    a[2, 0] = 0.0
    return a


def test_euler2mat_and_pose_vec2mat_bit_exact(dev):
    z, y, x = _angles(9, 1), _angles(9, 2).flip(0), _angles(9, 3)
    ref = R.euler2mat(z, y, x)
    got = sfm_amd.euler2mat(z.to(dev), y.to(dev), x.to(dev))
    assert got.shape == (9, 1, 3, 3) and torch.equal(got.cpu(), ref)
    vec = torch.cat([torch.randn(9, 3, generator=torch.Generator().manual_seed(4)), x, y, z], 1).contiguous()
    assert torch.equal(sfm_amd.pose_vec2mat(vec.to(dev)).cpu(), R.pose_vec2mat(vec))


@pytest.mark.parametrize("B,H,W,homog", [(2, 13, 17, True), (1, 128, 416, True), (3, 7, 1, False), (1, 1, 5, True)])
def test_meshgrid_bit_exact(dev, B, H, W, homog):
    got = sfm_amd.meshgrid(B, H, W, homog, device=dev)
    assert torch.equal(got.cpu(), R.meshgrid(B, H, W, homog))


@pytest.mark.parametrize("B,H,W,add_trans,homog", [(2, 16, 24, True, True), (3, 13, 17, False, True), (1, 37, 53, True, False)])
def test_pixel2cam_cam2pixel_bit_exact_and_compose_to_project_coords(dev, B, H, W, add_trans, homog):
    inp = synth.make_inputs(B, H, W, S=1, N=2, seed=77, pose_scale=0.05)
    depth, K, pose = inp["depth"][0].squeeze(3).contiguous(), inp["K"], inp["pose"][:, 0].contiguous()
    t = inp["trans"][0][..., :3].contiguous() if add_trans else None
    pix = R.meshgrid(B, H, W)
    ref_cam = R.pixel2cam(depth, pix, K, homog, add_trans, t)
    got_cam = sfm_amd.pixel2cam(depth.to(dev), pix.to(dev), K.to(dev), homog, add_trans, None if t is None else t.to(dev))
    assert torch.equal(got_cam.cpu(), ref_cam)
    if not homog:
        return
    proj = R._compose_proj(pose, K, False)
    ref_xy, ref_z = R.cam2pixel(ref_cam, proj)
    got_xy, got_z = sfm_amd.cam2pixel(got_cam, proj.to(dev))
    assert torch.equal(got_xy.cpu(), ref_xy.contiguous()) and torch.equal(got_z.cpu(), ref_z.contiguous())
    # the reference's composition (utils.py:185-202) from the pieces == the fused geometry kernel, bit for bit
    fx, fz = sfm_amd.project_coords(depth.to(dev), pose.to(dev), K.to(dev), False, add_trans, None if t is None else t.to(dev))
    assert torch.equal(fx, got_xy) and torch.equal(fz, got_z)


def test_helper_gradients_match_the_oracle(dev):
    B, H, W = 2, 9, 11
    inp = synth.make_inputs(B, H, W, S=1, N=2, seed=78, pose_scale=0.05)
    K = inp["K"]
    g = torch.Generator().manual_seed(5)
    wt = torch.randn(B, H, W, 2, generator=g)
    wz = torch.randn(B, H, W, 1, generator=g)

    def run(mod, to):
        vec = to(inp["pose"][:, 0].contiguous()).requires_grad_(True)
        depth = to(inp["depth"][0].squeeze(3).contiguous()).requires_grad_(True)
        t = to(inp["trans"][0][..., :3].contiguous()).requires_grad_(True)
        k = to(K)
        T = mod.pose_vec2mat(vec)
        filler = to(torch.tensor([0.0, 0.0, 0.0, 1.0]).reshape(1, 1, 4).repeat(B, 1, 1))
        k4 = torch.cat([torch.cat([k, to(torch.zeros(B, 3, 1))], 2), filler], 1)
        proj = torch.matmul(k4, T)
        pix = mod.meshgrid(B, H, W) if mod is R else mod.meshgrid(B, H, W, device=depth.device)
        cam = mod.pixel2cam(depth, pix, k, True, True, t)
        xy, z = mod.cam2pixel(cam, proj)
        loss = (xy * to(wt)).sum() + (z * to(wz)).sum()
        loss.backward()
        return vec.grad, depth.grad, t.grad

    ref = run(R, lambda v: v)
    got = run(sfm_amd, lambda v: v.to(dev))
    for a, b, name in zip(got, ref, ("d_vec", "d_depth", "d_trans")):
        assert rel_err(a, b) <= TOL_GRAD, name


@pytest.mark.parametrize("B,H,W,add_trans,homog", [(2, 9, 11, True, True), (1, 17, 40, False, True), (3, 5, 7, True, True)])
def test_every_helper_adjoint_is_hand_written_and_matches_the_oracle(dev, B, H, W, add_trans, homog):
    """csrc/geom.hip's adjoints one by one against autograd of the oracle: euler2mat (angles beyond +-pi included: the clip
    stops the gradient), pose_vec2mat, pixel2cam with respect to depth, pixel grid, INTRINSICS (through the matrix
    inverse) and translations, cam2pixel with respect to the camera points and the projection matrix - with both
    outputs used, and with only one of them"""
    g = torch.Generator().manual_seed(11)
    inp = synth.make_inputs(B, H, W, S=1, N=2, seed=79, pose_scale=0.05)
    ang = [torch.randn(B, 1, generator=g) * 2.5 for _ in range(3)]   # some beyond pi
    wR = torch.randn(B, 1, 3, 3, generator=g)
    wT = torch.randn(B, 4, 4, generator=g)

    def leaves(to):
        return dict(z=to(ang[0]).requires_grad_(True), y=to(ang[1]).requires_grad_(True), x=to(ang[2]).requires_grad_(True),
                    vec=to(inp["pose"][:, 1].contiguous()).requires_grad_(True),
                    depth=to(inp["depth"][0].squeeze(3).contiguous()).requires_grad_(True),
                    K=to(inp["K"].clone()).requires_grad_(True),
                    t=to(inp["trans"][0][..., 3:6].contiguous()).requires_grad_(True))

    for use in ("both", "coords", "z"):
        res = []
        for mod, to in ((R, lambda v: v.detach().clone()), (sfm_amd, lambda v: v.detach().to(dev))):
            L = leaves(to)
            pix = (mod.meshgrid(B, H, W) if mod is R else mod.meshgrid(B, H, W, device=dev)).requires_grad_(True)
            rot = mod.euler2mat(L["z"], L["y"], L["x"])
            T = mod.pose_vec2mat(L["vec"])
            cam = mod.pixel2cam(L["depth"], pix, L["K"], homog, add_trans, L["t"] if add_trans else None)
            proj = T * to(wT) + to(wT)      # a dense [B,4,4] that depends on vec
            xy, zz = mod.cam2pixel(cam, proj)
            loss = (rot * to(wR)).sum()
            if use in ("both", "coords"):
                loss = loss + (xy * xy).sum() * 1e-3
            if use in ("both", "z"):
                loss = loss + (zz * zz).sum()
            loss.backward()
            out = {k: v.grad for k, v in L.items() if (k != "t" or add_trans)}
            out["pix"] = pix.grad
            res.append(out)
        ref, got = res
        for k in ref:
            assert got[k] is not None, (k, use)
            assert rel_err(got[k], ref[k]) <= TOL_GRAD, (k, use, rel_err(got[k], ref[k]))


def test_helpers_reject_cpu_tensors():
    with pytest.raises(TypeError):
        sfm_amd.pose_vec2mat(torch.zeros(2, 6))
    with pytest.raises(TypeError):
        sfm_amd.meshgrid(1, 4, 4, device="cpu")
