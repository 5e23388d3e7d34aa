"""GPU parity of the fused multi-scale loss (csrc/loss.hip) against the CPU oracle's
multi_scale_loss (SfMLearner_uni.py:75-176 restated) and against the unfused composition
of the stand-alone HIP operators."""
import pytest
import torch

import sfm_amd
from sfm_amd import synth
from oracle import ref_graph as R
from helpers import TOL_GRAD, TOL_OUT, grad_close, min_select_ambiguous, rel_err

pytestmark = pytest.mark.gpu


def _oracle(inp, smooth_weight=0.0):
    depth = [d.clone().requires_grad_(True) for d in inp["depth"]]
    trans = [t.clone().requires_grad_(True) for t in inp["trans"]]
    pose = inp["pose"].clone().requires_grad_(True)
    loss = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], depth, pose, trans, pred_disp=inp["disp"],
                              smooth_weight=smooth_weight)
    loss.backward()
    return loss, depth, pose, trans


def _gpu(inp, dev, fused, smooth_weight=0.0):
    g = synth.to_device(inp, dev)
    depth = [d.requires_grad_(True) for d in g["depth"]]
    trans = [t.requires_grad_(True) for t in g["trans"]]
    pose = g["pose"].requires_grad_(True)
    loss = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], depth, pose, trans, pred_disp=g["disp"],
                                    smooth_weight=smooth_weight, fused=fused)
    loss.backward()
    return loss, depth, pose, trans


@pytest.mark.parametrize("B,H,W,S,N,ps,full", [
    (2, 16, 24, 2, 2, 0.01, True),
    (1, 37, 53, 3, 2, 0.05, True),     # ragged: tiles and blocks with tails
    (2, 32, 64, 4, 2, 0.01, False),    # pyramid inputs, resized as SfMLearner_uni.py:91,113
    (1, 24, 40, 2, 4, 0.02, True),     # seq=5: sources 2,3 only feed the splat terms (quirk 3)
    (2, 20, 36, 2, 3, 0.02, True),     # three sources: the backward's second launch handles a single one
    (1, 128, 416, 4, 2, 0.01, True),   # one KITTI snippet, 4 scales
    (1, 384, 1248, 1, 2, 0.01, True),  # one full-resolution KITTI snippet (configs[2] image size): beyond one strip chunk
                                       # per wave, 20 x 12 splat tiles, the longest rows the oracle affords in seconds
    (8, 128, 416, 4, 2, 0.01, True),   # configs[1] at its FULL per-GPU size against the oracle (the GPU box's host cores
                                       # run the oracle's forward + backward at this size in about a second)
    (4, 384, 1248, 4, 2, 0.01, True),  # configs[2]: the 8-GPU shard (B = 32 / 8) at full size, all four scales
])
@pytest.mark.parametrize("fused", [True, False])
def test_multi_scale_loss(dev, B, H, W, S, N, ps, full, fused):
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=31, pose_scale=ps, full_res=full)
    lo, do, po, to = _oracle(inp)
    lg, dg, pg, tg = _gpu(inp, dev, fused)
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(pg.grad, po.grad) <= TOL_GRAD
    g = synth.to_device(inp, dev)
    for s in range(S):
        # per-pixel gradients: strict max-norm outside the 3x3 neighbourhoods of the pixels whose
        # min-reprojection selection is below fp32 resolution (helpers.min_select_ambiguous)
        with torch.no_grad():
            depth_full = sfm_amd.loss.resize_bilinear(g["depth"][s], H, W).squeeze(3).contiguous()
            trans_full = sfm_amd.loss.resize_bilinear(g["trans"][s], H, W).contiguous()
            amb = min_select_ambiguous(g["tgt"], g["src"], g["K"], depth_full, g["pose"], trans_full)
        if not full:  # gradients live at the pyramid resolution: any ambiguous pixel may touch any coarse cell
            amb = None if not bool(amb.any()) else "skip"
        for name, a, b in (("d_depth", dg[s].grad, do[s].grad), ("d_trans", tg[s].grad, to[s].grad)):
            if amb is None or isinstance(amb, str):
                ok, msg = grad_close(a, b, max_outlier_frac=0.0 if amb is None else 2e-3)
            else:
                # strict below a million pixels; beyond, a handful of elements (4 of 11.5 M d_trans values fused, 9 of
                # 1.9 M d_depth values unfused at configs[2]) sit behind one of the other discontinuities - sign(), == 0,
                # floor - decided differently by two fp32 evaluation orders: grad_close's default allowance of 1e-5
                ok, msg = grad_close(a, b, max_outlier_frac=0.0 if B * H * W < 1_000_000 else 1e-5, exclude=amb)
            assert ok, f"{name} scale {s}: {msg}"


@pytest.mark.timeout(900)
def test_config4_snippet_against_the_oracle(dev):
    """configs[3] (2048 x 1024, seq = 5, Cityscapes intrinsics) at full image size against the oracle: one snippet, all
    four scales and four sources - the overflow path of the splat (large parallax) and the second backward launch
    (sources 2, 3) at the size they are meant for"""
    B, H, W, S, N = 1, 1024, 2048, 4, 4
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=33, pose_scale=0.01, full_res=True, cityscapes=True)
    lo, do, po, to = _oracle(inp)
    lg, dg, pg, tg = _gpu(inp, dev, True)
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(pg.grad, po.grad) <= 3 * TOL_GRAD   # (sums every pixel, the min-select-ambiguous ones included)
    g = synth.to_device(inp, dev)
    for s in range(S):
        with torch.no_grad():
            amb = min_select_ambiguous(g["tgt"], g["src"], g["K"], g["depth"][s].squeeze(3).contiguous(), g["pose"],
                                       g["trans"][s].contiguous())
        for name, a, b in (("d_depth", dg[s].grad, do[s].grad), ("d_trans", tg[s].grad, to[s].grad)):
            ok, msg = grad_close(a, b, exclude=amb)
            assert ok, f"{name} scale {s}: {msg}"
    print(f"c4 snippet: loss rel err {rel_err(lg, lo):.2e}, d_pose rel err {rel_err(pg.grad, po.grad):.2e}")


def test_loss_with_smoothness_terms(dev):
    inp = synth.make_inputs(2, 32, 64, S=3, N=2, seed=32, full_res=False)
    lo, do, po, to = _oracle(inp, smooth_weight=0.01)
    lg, dg, pg, tg = _gpu(inp, dev, True, smooth_weight=0.01)
    assert rel_err(lg, lo) <= TOL_OUT
    for s in range(3):
        assert rel_err(tg[s].grad, to[s].grad) <= TOL_GRAD


def test_upstream_gradient_scales(dev):
    inp = synth.make_inputs(1, 16, 24, S=2, N=2, seed=33)
    g = synth.to_device(inp, dev)
    pose = g["pose"].requires_grad_(True)
    loss = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], g["depth"], pose, g["trans"])
    (3.0 * loss).backward()
    g3 = pose.grad.clone()
    pose.grad = None
    loss = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], g["depth"], pose, g["trans"])
    loss.backward()
    assert rel_err(g3, 3.0 * pose.grad) <= 1e-6


def test_fused_is_deterministic_up_to_atomics(dev):
    inp = synth.make_inputs(2, 32, 64, S=2, N=2, seed=34)
    a = _gpu(inp, dev, True)
    b = _gpu(inp, dev, True)
    assert rel_err(a[0], b[0]) <= 1e-6
    assert rel_err(a[2].grad, b[2].grad) <= 1e-5


def test_overflow_path_and_workspace_invariant(dev):
    """Large poses push many splat corners outside the scan windows of the tile-owned forward
    splat (csrc/loss.hip): they take the global-atomic overflow path.  Steps that share one
    workspace must not see each other's overflow: A (overflow, forward only) -> B (overflow,
    with gradients) -> C (small pose), each against the oracle."""
    B, H, W, S, N = 1, 96, 200, 2, 2
    cases = [(61, 0.15, False), (62, 0.15, True), (63, 0.01, True)]
    for seed, ps, with_grad in cases:
        inp = synth.make_inputs(B, H, W, S=S, N=N, seed=seed, pose_scale=ps)
        lo, do, po, to = _oracle(inp)
        if with_grad:
            lg, dg, pg, tg = _gpu(inp, dev, True)
            assert rel_err(pg.grad, po.grad) <= TOL_GRAD
            for s in range(S):
                assert rel_err(dg[s].grad, do[s].grad) <= TOL_GRAD
                assert rel_err(tg[s].grad, to[s].grad) <= TOL_GRAD
        else:
            g = synth.to_device(inp, dev)
            with torch.no_grad():
                lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], g["depth"], g["pose"], g["trans"])
        assert rel_err(lg, lo) <= TOL_OUT, f"seed {seed}"
    eng = sfm_amd.loss.get_engine(B, H, W, S, N, dev)
    torch.cuda.synchronize()
    assert eng.overflow_state() == (0, 0.0), "Ovf / flags not restored to zero after the steps"


def _stacked_dev(inp, dev):
    g = synth.to_device(inp, dev)
    depth = torch.stack([d.squeeze(3) for d in g["depth"]], 0).contiguous()
    trans = torch.stack(g["trans"], 0).contiguous()
    return g, depth, trans


@pytest.mark.parametrize("B,H,W,S,N,ps", [(2, 32, 64, 2, 2, 0.01), (1, 96, 200, 2, 2, 0.15), (1, 37, 53, 2, 4, 0.3),
                                          (1, 128, 416, 2, 2, 0.1)])
def test_mask_sums_taken_before_the_stencil_equal_the_stencils_own(dev, B, H, W, S, N, ps):
    """kV0 of the combined seed plane uses M(s, i) = sum_q nz(q) wgt(q), counted where the splat tiles are written and
    corrected for the pixels that received overflow contributions (large poses: most tiles do); the stencil's own
    sum of m = avgpool3x3(nz) over the finished accumulators (partials[..., 1]) must agree, and so must the oracle's."""
    from sfm_amd.loss import LossEngine
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=71, pose_scale=ps)
    g, depth, trans = _stacked_dev(inp, dev)
    eng = LossEngine(B, H, W, S, N, dev)
    ne = B * H * W * 3
    for _ in range(2):
        # the three-call forward hands the mask sums out (the single-call one keeps the tile shares in the workspace)
        early = eng.forward_warps(g["tgt"], g["src"], g["K"], depth, g["pose"], trans)
        eng.forward_gather(g["tgt"], ne, True)
        p = torch.empty(eng.n_partials, dtype=torch.float64, device=dev)
        eng.forward_splat(g["src"], ne, True, early, p)
        torch.cuda.synchronize()
        n_flagged, _ = eng.overflow_state()
        eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, p, ne, 1.0, True, seeds_guarded=False)
        # ... and the single-call forward (fix-up inside the gather-side launch, tile shares added up by the stencil
        # waves) gives the same sums and gradients bit for bit
        q = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
        if n_flagged == 0:
            assert torch.equal(q, p)
        else:  # overflow contributions are global float atomics: their order, hence the last bits, change from run to run
            assert rel_err(q, p) <= 1e-6
        eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, q, ne, 1.0, True)
        late = p.view(S, 3 * N + 1)[:, 1:3 * N:3].reshape(-1)
        assert rel_err(early, late) <= 1e-6, f"{n_flagged} tiles with overflow"
    if ps >= 0.1:
        assert n_flagged > 0, "the case was meant to exercise the overflow fix-up"
    ref = R.loss_partials(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"]).view(S, 3 * N + 1)
    assert rel_err(early.cpu(), ref[:, 1:3 * N:3].reshape(-1)) <= 1e-5


def test_seed_guard_rewrites_the_seeds_when_the_clip_does_not_pass(dev):
    """clip((1 - mean ssim) / 2, 0, 1) (utils.py:575) has no gradient outside [0, 1].  The seeds are written before the
    global mean SSIM exists, assuming it passes; when the reduced sums say otherwise the backward first rewrites the
    seeds of that (scale, source) with kV0 = 0.  Forced here by handing the backward sums with mean SSIM = 1.5 for some
    pairs; reference = autograd of the oracle's loss_from_partials on the same sums."""
    from sfm_amd.loss import LossEngine
    B, H, W, S, N = 2, 32, 64, 2, 2
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=72, pose_scale=0.02)
    g, depth, trans = _stacked_dev(inp, dev)
    eng = LossEngine(B, H, W, S, N, dev)
    ne = B * H * W * 3
    for failing in ([(0, 1)], [(0, 0), (1, 0), (1, 1)], []):
        p = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
        for s, i in failing:
            p.view(S, 3 * N + 1)[s, 3 * i + 2] = 1.5 * ne
        loss, d_depth, d_pose, d_trans = eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, p, ne, 1.0, True,
                                                      seeds_guarded=False)   # the sums were edited after the forward
        do = [d.clone().requires_grad_(True) for d in inp["depth"]]
        to = [t.clone().requires_grad_(True) for t in inp["trans"]]
        po = inp["pose"].clone().requires_grad_(True)
        part = R.loss_partials(inp["tgt"], inp["src"], inp["K"], do, po, to)
        fake = part.detach().clone().view(S, 3 * N + 1)
        for s, i in failing:
            fake[s, 3 * i + 2] = 1.5 * ne
        lo = R.loss_from_partials(part + (fake.reshape(-1) - part.detach()), S, N, ne)
        lo.backward()
        assert rel_err(loss, lo.float()) <= TOL_OUT
        assert rel_err(d_pose, po.grad) <= TOL_GRAD, failing
        for s in range(S):
            assert rel_err(d_depth[s], do[s].grad.squeeze(3)) <= TOL_GRAD, failing
            assert rel_err(d_trans[s], to[s].grad) <= TOL_GRAD, failing


def test_mismatched_pyramid_maps_are_rejected_or_resized_separately(dev):
    """the pyramid kernels index both maps of a level with ONE size: a depth map sized unlike its translation map
    (legal in the reference, which resizes each on its own) must take the materialised resize; wrong batch / channel
    counts must raise before anything is launched"""
    B, H, W, S, N = 2, 32, 64, 2, 2
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=73, full_res=False)
    odd = [inp["depth"][0], torch.nn.functional.interpolate(inp["depth"][1].permute(0, 3, 1, 2), size=(12, 20)).permute(0, 2, 3, 1).contiguous()]
    lo = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], odd, inp["pose"], inp["trans"])
    g = synth.to_device(inp, dev)
    lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], [d.to(dev) for d in odd], g["pose"], g["trans"])
    assert rel_err(lg, lo) <= TOL_OUT
    with pytest.raises(ValueError, match="do not match"):
        sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], [g["depth"][0][:1], g["depth"][1]], g["pose"], g["trans"])
    with pytest.raises(ValueError, match="do not match"):
        sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], g["depth"], g["pose"], [g["trans"][0][..., :3], g["trans"][1]])


def test_single_device_forward_applies_the_seed_guard_itself(dev):
    """the same guard inside sfm_loss_forward's reduction (single device): forced with images whose splatted view equals
    the source exactly - constant images - so that the mean SSIM is 1 up to rounding and the clip is at its edge; whatever
    side the rounding falls on, the gradients must agree with the guard applied by the backward (need_grad = 1 path)"""
    from sfm_amd.loss import LossEngine
    B, H, W, S, N = 1, 32, 64, 2, 2
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=74, pose_scale=0.02)
    inp["tgt"] = torch.full_like(inp["tgt"], 0.37)
    inp["src"] = torch.full_like(inp["src"], 0.37)
    g, depth, trans = _stacked_dev(inp, dev)
    eng = LossEngine(B, H, W, S, N, dev)
    ne = B * H * W * 3
    p = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
    a = eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, p, ne, 1.0, True)                       # guarded by the forward
    p2 = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
    b = eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, p2, ne, 1.0, True, seeds_guarded=False)  # guard again: idempotent
    assert torch.equal(p, p2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    v = p.view(S, 3 * N + 1)[:, 2:3 * N:3] / ne
    print("mean SSIM per (scale, source):", v.flatten().tolist())
