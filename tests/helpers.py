"""Shared helpers of the parity tests.  Tolerances are BASELINE.json's: outputs and losses
1e-5 relative, gradients 1e-4 relative (atomic re-ordering), integer indices bit-exact.
"Relative" is tensor-level: max|a-b| / max(max|b|, floor)."""
import torch

from sfm_amd import synth

TOL_OUT = 1e-5
TOL_GRAD = 1e-4


def rel_err(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-6) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def warp_case(B, H, W, seed=0, pose_scale=0.01, add_trans=True):
    """One (img_src, img_tgt, depth, pose, K, trans) warp problem on the CPU."""
    inp = synth.make_inputs(B, H, W, S=1, N=2, seed=seed, pose_scale=pose_scale)
    return dict(src=inp["src"][..., 0:3].contiguous(), tgt=inp["tgt"], depth=inp["depth"][0].squeeze(3).contiguous(),
                pose=inp["pose"][:, 0].contiguous(), K=inp["K"],
                trans=inp["trans"][0][..., 0:3].contiguous() if add_trans else None)


CASES = [  # (B, H, W, pose_scale): tiny, ragged (no multiple of anything), KITTI row, stress (out of bounds)
    (2, 16, 24, 0.01),
    (3, 13, 17, 0.02),
    (1, 128, 416, 0.01),
    (2, 37, 53, 0.1),
]


def grad_close(a: torch.Tensor, b: torch.Tensor, tol: float = TOL_GRAD, max_outlier_frac: float = 1e-5):
    """Per-pixel gradients at full image sizes: the loss has genuine discontinuities (the per-pixel
    min of SfMLearner_uni.py:168, sign(), the clips), so two fp32 evaluation orders may pick different
    branches at a handful of near-tie pixels.  Required: relative L2 error <= tol and at most
    max_outlier_frac of the elements off by more than tol * max|b|.  Returns (ok, message)."""
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    scale = max(float(b.abs().max()), 1e-30)
    l2 = float((a - b).norm() / max(float(b.norm()), 1e-30))
    frac = float(((a - b).abs() > tol * scale).double().mean())
    return (l2 <= tol and frac <= max_outlier_frac), f"rel-L2 {l2:.2e}, outliers {frac:.2e}, max {rel_err(a, b):.2e}"
