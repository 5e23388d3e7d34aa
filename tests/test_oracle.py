"""CPU pins of the oracle (oracle/ref_graph.py).  The reference ships no tests or fixtures
and TensorFlow is not installable here, so parity with the reference itself is UNPINNED;
what pins the restatement is (1) analytic known answers derived from the reference's
formulas, (2) an independent per-pixel NumPy restatement, (3) fp64 gradcheck of the
differentiable pieces and (4) the committed golden vectors (drift guard)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import np_forward as NP
from oracle import ref_graph as R
from sfm_amd import synth
from helpers import rel_err, warp_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _K(B, fx=20.0, fy=22.0, cx=11.5, cy=7.5):
    return torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])[None].repeat(B, 1, 1)


# ------------------------------------------------------------------ (1) known answers
def test_meshgrid_is_not_exact_integers_quirk5():
    """utils.py:157-163 through fp32 linspace: SURVEY.md quirk 5 measured 136/416 inexact columns at W=416,
    27 of them below the integer, max |x - i| = 3.05e-5."""
    g = R.meshgrid(1, 128, 416)[0, 0, 0]
    i = torch.arange(416, dtype=torch.float32)
    assert int((g != i).sum()) == 136
    assert int((g < i).sum()) == 27
    assert abs(float((g - i).abs().max()) - 3.05e-5) < 1e-6


def test_euler_clip_and_composition():
    # angles beyond +-pi are clipped (utils.py:42-44): rz = 4 behaves as rz = pi
    a = R.pose_vec2mat(torch.tensor([[0, 0, 0, 0, 0, 4.0]]))
    b = R.pose_vec2mat(torch.tensor([[0, 0, 0, 0, 0, math.pi]]))
    assert torch.equal(a, b)
    # R = X . Y . Z (utils.py:74) against a direct fp64 product
    v = torch.tensor([[0.1, -0.2, 0.3, 0.3, -0.2, 0.5]])
    rx, ry, rz = 0.3, -0.2, 0.5
    X = np.array([[1, 0, 0], [0, math.cos(rx), -math.sin(rx)], [0, math.sin(rx), math.cos(rx)]])
    Y = np.array([[math.cos(ry), 0, math.sin(ry)], [0, 1, 0], [-math.sin(ry), 0, math.cos(ry)]])
    Z = np.array([[math.cos(rz), -math.sin(rz), 0], [math.sin(rz), math.cos(rz), 0], [0, 0, 1]])
    T = R.pose_vec2mat(v)[0].double().numpy()
    assert np.allclose(T[:3, :3], X @ Y @ Z, atol=1e-6)
    assert np.allclose(T[:3, 3], [0.1, -0.2, 0.3]) and np.allclose(T[3], [0, 0, 0, 1])


def test_matrix_inverse_lu_matches_linalg():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(5, 4, 4, generator=g, dtype=torch.float64) + 3 * torch.eye(4, dtype=torch.float64)
    assert torch.allclose(R.matrix_inverse_lu(a), torch.linalg.inv(a), atol=1e-10)


def test_identity_pose_gather_returns_image_up_to_quirks():
    """identity pose, unit depth, no translations: coords == the (inexact) meshgrid, so the gather returns
    the image where the grid is an exact integer < W-1 and the clipped-corner quirk 6 zeroes the last row/col."""
    B, H, W = 1, 8, 12
    img = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1))
    out, mask = R.projective_inverse_warp(img.double(), torch.ones(B, H, W, dtype=torch.float64),
                                          torch.zeros(B, 6, dtype=torch.float64), _K(B).double())
    assert torch.allclose(out[:, :-1, :-1], img.double()[:, :-1, :-1], atol=1e-9)
    assert torch.equal(mask, torch.ones(B, H, W, 1, dtype=torch.float64))
    # exactly on the last row / column both clipped corners coincide and both weights are 0 (quirk 6)
    base = torch.stack(torch.meshgrid(torch.arange(W, dtype=torch.float64), torch.arange(H, dtype=torch.float64),
                                      indexing="xy"), dim=-1)[None]
    out, _ = R.bilinear_sampler(img.double(), base)
    assert torch.equal(out[:, :-1, :-1], img.double()[:, :-1, :-1])
    assert float(out[:, -1].abs().max()) == 0.0 and float(out[:, :, -1].abs().max()) == 0.0


def test_integer_x_translation_shifts_the_gather():
    """tx = k*d/fx moves every projected x by exactly k pixels (z unchanged)."""
    B, H, W, k = 1, 8, 12, 3
    Kd = _K(B).double()
    img = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    depth = torch.full((B, H, W), 2.0, dtype=torch.float64)
    pose = torch.zeros(B, 6, dtype=torch.float64)
    pose[:, 0] = k * 2.0 / 20.0
    out, _ = R.projective_inverse_warp(img, depth, pose, Kd)
    assert torch.allclose(out[:, :-1, :W - k - 1], img[:, :-1, k:W - 1], atol=1e-9)


def test_splat_of_constant_is_the_constant_and_identity_blur():
    B, H, W = 1, 8, 12
    c = warp_case(B, H, W, seed=3, pose_scale=0.05)
    const = torch.full((B, H, W, 3), 0.37)
    out, mask = R.projective_warp(const, c["depth"], c["pose"], c["K"], add_trans=True, translations=c["trans"])
    hit = out != 0
    assert hit.any() and float((out[hit] - 0.37).abs().max()) < 1e-6
    # identity pose (fp64, exact grid): every pixel lands on itself; the four one-sided Gaussian
    # weights are (1, e^-1/2, e^-1/2, e^-1) x exp(-z/4) (utils.py:431-442)
    img = torch.rand(B, H, W, 1, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    base = torch.stack(torch.meshgrid(torch.arange(W, dtype=torch.float64), torch.arange(H, dtype=torch.float64),
                                      indexing="xy"), dim=-1)[None]
    out, _ = R.bilinear_project(img, base, use_depth=True, depth=torch.full((B, H, W, 1), 2.0, dtype=torch.float64))
    e = math.exp(-0.5)
    i = img[0, :, :, 0]
    v, u = 3, 5  # interior pixel: receives from itself (1), left (e), up (e), up-left (e*e)
    num = i[v, u] + e * i[v, u - 1] + e * i[v - 1, u] + e * e * i[v - 1, u - 1]
    den = 1 + 2 * e + e * e
    assert abs(float(out[0, v, u, 0]) - float(num / den)) < 1e-9   # exp(-z/4) cancels in the ratio


def test_sampler_out_of_bounds_cancels_quirk6():
    img = torch.rand(1, 6, 7, 3, generator=torch.Generator().manual_seed(5))
    coords = torch.tensor([[[[-5.3, 2.2], [9.7, 2.2], [3.1, -4.0], [3.1, 11.0], [6.0, 5.0]]]])
    out, mask = R.bilinear_sampler(img, coords)
    assert float(out.abs().max()) < 1e-6
    assert mask.flatten().tolist() == [0.0, 0.0, 0.0, 0.0, 1.0]


def test_ssim_identical_images_is_one_and_window_normalised():
    w = R.fspecial_gauss(3, 1.5)
    assert abs(float(w.sum()) - 1) < 1e-6 and abs(float(w[1, 1] / w[0, 1]) - math.exp(1 / 4.5)) < 1e-5
    a = torch.rand(1, 9, 9, 3, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    assert torch.allclose(R.ssim_value(a, a), torch.ones_like(a), atol=1e-12)
    assert float(R.tf_ssim(a, a)) == 0.0
    # two constant images p, q: zero variance, so value = (2pq + C1) / (p^2 + q^2 + C1) at interior pixels
    p_, q_ = 0.3, -0.6
    v = R.ssim_value(torch.full((1, 5, 5, 3), p_, dtype=torch.float64), torch.full((1, 5, 5, 3), q_, dtype=torch.float64))
    assert abs(float(v[0, 2, 2, 0]) - (2 * p_ * q_ + 1e-4) / (p_ * p_ + q_ * q_ + 1e-4)) < 1e-9


def test_float_index_quirk4_equals_integer_index_below_2_24():
    c = warp_case(2, 16, 24, seed=7, pose_scale=0.05)
    a, _ = R.projective_inverse_warp(c["src"], c["depth"], c["pose"], c["K"], float_index=True)
    b, _ = R.projective_inverse_warp(c["src"], c["depth"], c["pose"], c["K"], float_index=False)
    assert torch.equal(a, b)


def test_resize_bilinear_tf1_semantics():
    x = torch.arange(6.0).reshape(1, 2, 3, 1)
    y = R.resize_bilinear_tf1(x, 4, 6)   # src = dst * in/out, top-left anchored, clamped at the far edge
    assert y[0, 0, :, 0].tolist() == [0.0, 0.5, 1.0, 1.5, 2.0, 2.0]
    assert y[0, :, 0, 0].tolist() == [0.0, 1.5, 3.0, 3.0]


def test_reference_crashes_are_kept():
    inp = synth.make_inputs(1, 8, 12, S=1, N=2, seed=8)
    with pytest.raises(NameError):   # SfMLearner_uni.py:127 with explain_reg_weight == 0
        R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"], explain_reg_weight=0)
    with pytest.raises(IndexError):  # curr_errors[1], SfMLearner_uni.py:168
        R.multi_scale_loss(inp["tgt"], inp["src"][..., :3], inp["K"], inp["depth"], inp["pose"][:, :1],
                           [t[..., :3] for t in inp["trans"]])


# ------------------------------------------------------------------ (2) independent restatement
@pytest.mark.parametrize("B,H,W,ps,inv", [(2, 9, 13, 0.05, False), (1, 7, 10, 0.3, False), (1, 6, 9, 0.05, True)])
def test_numpy_restatement_agrees(B, H, W, ps, inv):
    c = warp_case(B, H, W, seed=9, pose_scale=ps)
    n = {k: v.numpy() for k, v in c.items()}
    coords_t, z_t = R.project_coords(c["depth"], c["pose"], c["K"], inv, True, c["trans"])
    coords_n, z_n = NP.project_coords(n["depth"], n["pose"], n["K"], inv, n["trans"])
    assert np.array_equal(coords_n, coords_t.numpy())          # same canonical fp32 op order: bitwise
    assert np.array_equal(z_n, z_t.numpy())
    out_t, mask_t = R.bilinear_sampler(c["src"], coords_t)
    out_n, mask_n, idx_n = NP.bilinear_sampler(n["src"], coords_n)
    _, ints = R.corner_indices(coords_t, H, W)
    assert np.array_equal(idx_n, torch.cat(ints, dim=3).numpy())
    assert np.array_equal(out_n, out_t.numpy()) and np.array_equal(mask_n, mask_t.numpy())
    sp_t, m_t = R.bilinear_project(c["tgt"], coords_t, use_depth=True, depth=z_t)
    sp_n, m_n = NP.bilinear_project(n["tgt"], coords_n, z_n)
    assert rel_err(torch.from_numpy(sp_n), sp_t) < 1e-5
    assert np.array_equal(m_n != 0, m_t.numpy() != 0) and rel_err(torch.from_numpy(m_n), m_t) < 1e-6


# ------------------------------------------------------------------ (3) gradcheck (fp64)
def test_gradcheck_warps_and_ssim():
    c = warp_case(1, 5, 6, seed=10, pose_scale=0.05)
    d = {k: v.double() for k, v in c.items()}
    depth, pose, trans = (d[k].clone().requires_grad_(True) for k in ("depth", "pose", "trans"))
    assert torch.autograd.gradcheck(
        lambda a, b, t: R.projective_inverse_warp(d["src"], a, b, d["K"], add_trans=True, translations=t)[0],
        (depth, pose, trans), eps=1e-7, atol=1e-5, nondet_tol=1e-12)
    assert torch.autograd.gradcheck(
        lambda a, b, t: R.projective_warp(d["tgt"], a, b, d["K"], add_trans=True, translations=t)[0],
        (depth, pose, trans), eps=1e-7, atol=1e-5, nondet_tol=1e-12)
    x = torch.rand(1, 5, 5, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(11)).requires_grad_(True)
    y = torch.rand(1, 5, 5, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(12)).requires_grad_(True)
    assert torch.autograd.gradcheck(R.ssim_value, (x, y), eps=1e-7, atol=1e-5)


# ------------------------------------------------------------------ (4) golden vectors (drift guard)
@pytest.mark.parametrize("name,B,H,W,seed,ps", [("warp_2x16x24", 2, 16, 24, 101, 0.05),
                                                 ("warp_3x13x17_oob", 3, 13, 17, 102, 0.2)])
def test_golden_warp(name, B, H, W, seed, ps):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    c = warp_case(B, H, W, seed=seed, pose_scale=ps)
    for k in ("src", "tgt", "depth", "pose", "K", "trans"):
        assert np.array_equal(g[k], c[k].numpy()), f"synthetic input {k} drifted"
    coords, z = R.project_coords(c["depth"], c["pose"], c["K"], False, True, c["trans"])
    _, ints = R.corner_indices(coords, H, W)
    assert np.array_equal(g["idx"], torch.cat(ints, dim=3).numpy())
    assert np.array_equal(g["coords"], coords.numpy())
    out, mask = R.projective_inverse_warp(c["src"], c["depth"], c["pose"], c["K"], add_trans=True, translations=c["trans"])
    assert np.array_equal(g["gather_out"], out.numpy()) and np.array_equal(g["gather_mask"], mask.numpy())
    out, mask = R.projective_warp(c["tgt"], c["depth"], c["pose"], c["K"], add_trans=True, translations=c["trans"])
    assert rel_err(out, torch.from_numpy(g["splat_out"])) < 1e-6     # index_add order may vary with threads
    assert rel_err(mask, torch.from_numpy(g["splat_mask"])) < 1e-6


def test_golden_loss_and_partials():
    g = np.load(os.path.join(GOLD, "loss_2x16x24_s2n2.npz"))
    B, H, W, S, N = (int(v) for v in g["dims"])
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=103, pose_scale=0.02, full_res=True)
    loss = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    part = R.loss_partials(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    assert rel_err(part, torch.from_numpy(g["partials"])) < 1e-6
    assert abs(float(R.loss_from_partials(part, S, N, B * H * W * 3)) - float(loss)) <= 2e-6 * abs(float(loss))


def test_golden_c1_config0():
    """BASELINE.json configs[0]: bilinear_project forward on one 416x128 snippet, B=1 (CPU path)."""
    g = np.load(os.path.join(GOLD, "c1_project_1x128x416.npz"))
    c = warp_case(1, 128, 416, seed=synth.SEED, pose_scale=0.01)
    sums = np.array([float(c[k].double().sum()) for k in ("tgt", "depth", "pose", "trans")])
    assert np.allclose(sums, g["input_sums"], rtol=1e-12)
    coords, z = R.project_coords(c["depth"], c["pose"], c["K"], False, True, c["trans"])
    _, ints = R.corner_indices(coords, 128, 416)
    assert np.array_equal(torch.cat(ints, dim=3).numpy().astype(np.int16), g["idx"])
    out, mask = R.bilinear_project(c["tgt"], coords, use_depth=True, depth=z)
    assert rel_err(out, torch.from_numpy(g["out"])) < 1e-6 and rel_err(mask, torch.from_numpy(g["mask"])) < 1e-6


# ------------------------------------------------------------------ (5) the pyramid steps and the ASPP epilogue
def test_resize_x2_known_answer():
    """TF-1 legacy resize (align_corners=False, no half-pixel centres), x2: even outputs copy, odd outputs average the
    two neighbours, the last one clamps."""
    x = torch.arange(12, dtype=torch.float32).reshape(1, 3, 4, 1) ** 2
    y = R.resize_bilinear_tf1(x, 6, 8)
    assert torch.equal(y[0, ::2, ::2, 0], x[0, :, :, 0])
    assert torch.equal(y[0, 0, 1::2, 0][:-1], (x[0, 0, :-1, 0] + x[0, 0, 1:, 0]) / 2)
    assert torch.equal(y[0, 0, -1, 0], x[0, 0, -1, 0]) and torch.equal(y[0, -1, 0, 0], x[0, -1, 0, 0])
    assert torch.equal(R.resize_bilinear_tf1(x, 3, 4), x)


def test_smooth_loss_known_answer():
    """p = a x^2 + b y: |dx| = a (2x + 1), dy = b, dx2 = 2a, every mixed / dy2 term 0."""
    a, b, h, w = 0.5, -0.25, 5, 7
    xs = torch.arange(w, dtype=torch.float32)
    p = (a * xs ** 2)[None, None, :, None] + (b * torch.arange(h, dtype=torch.float32))[None, :, None, None]
    p = p.expand(2, h, w, 3).contiguous()
    want = 0.25 * (float((a * (2 * xs[:-1] + 1)).mean()) + abs(b)) + 0.75 * (2 * a)
    assert abs(float(R.compute_smooth_loss(p)) - want) < 1e-6


def test_normalize_disp_known_answer():
    d = [torch.rand(2, 4, 6, 1) + 0.5, torch.rand(2, 2, 3, 1) + 0.5]
    dn, dp = R.normalize_disp(d)
    for u, v in zip(dn, dp):
        assert torch.allclose(u.mean(dim=(1, 2, 3)), torch.ones(2), atol=1e-6)
        assert torch.equal(v, 1.0 / u)


def test_batch_norm_elu_against_torch_functional():
    """independent implementation: F.batch_norm(training) + F.elu; momentum = 1 - decay; running_var is unbiased"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 7, 8, generator=g) * 2 + 1
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    mm, mv = torch.zeros(8), torch.ones(8)
    y = R.batch_norm_elu(x, gamma, beta, mm, mv, decay=0.9, epsilon=1e-5)
    rm, rv = torch.zeros(8), torch.ones(8)
    y2 = F.elu(F.batch_norm(x.permute(0, 3, 1, 2), rm, rv, gamma, beta, True, 0.1, 1e-5)).permute(0, 2, 3, 1)
    assert rel_err(y, y2) < 1e-5 and rel_err(mm, rm) < 1e-5 and rel_err(mv, rv) < 1e-5


def test_gradcheck_pyramid_ops():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 3, 4, 2, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: R.resize_bilinear_tf1(t, 7, 9), (x,), eps=1e-6, atol=1e-6)
    d = (torch.rand(2, 3, 4, 1, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda t: torch.cat([v.flatten() for v in sum(R.normalize_disp([t]), [])]), (d,),
                                    eps=1e-6, atol=1e-6)
    gam = (torch.rand(2, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    bet = torch.randn(2, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b, c: R.batch_norm_elu(a, b, c), (x, gam, bet), eps=1e-6, atol=1e-5)


def test_golden_pyramid_and_bn():
    g = np.load(os.path.join(GOLD, "pyramid_bn_2x16x24.npz"))
    B, H, W, S, N = (int(v) for v in g["dims"])
    raw = [torch.from_numpy(g[f"raw{s}"]) for s in range(S)]
    exp = [torch.from_numpy(g[f"exp{s}"]) for s in range(S)]
    dn, dp = R.normalize_disp(raw)
    for s in range(S):
        assert rel_err(dn[s], torch.from_numpy(g[f"dispn{s}"])) < 1e-6
        assert np.array_equal(R.resize_bilinear_tf1(torch.from_numpy(g[f"depth{s}"]), H, W).numpy(), g[f"up_depth{s}"])
        assert np.array_equal(R.resize_bilinear_tf1(exp[s], H, W).numpy(), g[f"up_exp{s}"])
    sm = sum(0.01 / 2 ** s * R.compute_smooth_loss(dn[s]) + 0.01 / 2 ** s * (R.compute_smooth_loss(exp[s]) + exp[s].abs().mean())
             for s in range(S))
    assert abs(float(sm) - float(g["smooth"])) <= 1e-6 * abs(float(g["smooth"]))
    y = R.batch_norm_elu(torch.from_numpy(g["bn_x"]), torch.from_numpy(g["bn_gamma"]), torch.from_numpy(g["bn_beta"]))
    assert rel_err(y, torch.from_numpy(g["bn_y"])) < 1e-6
