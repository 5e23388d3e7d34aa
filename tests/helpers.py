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


def grad_close(a: torch.Tensor, b: torch.Tensor, tol: float = TOL_GRAD, max_outlier_frac: float = 1e-5, exclude=None):
    """Per-pixel gradients at full image sizes: the loss has genuine discontinuities (the per-pixel
    min of SfMLearner_uni.py:168, sign(), the clips), so two fp32 evaluation orders may pick different
    branches at a handful of near-tie pixels.  ``exclude`` (bool, broadcastable to a's shape) marks the
    pixels whose min-selection is numerically undecided (``min_select_ambiguous``); they are left out.
    Required on the rest: at most max_outlier_frac of the elements off by more than tol * max|b|, and a
    relative L2 error <= tol over the others.  Returns (ok, message)."""
    if a.device != b.device:
        a, b = a.cpu(), b.cpu()
    a = a.detach().double()    # stays on the device when both are there (the full-size cases are GBs)
    b = b.detach().double()
    scale = max(float(b.abs().max()), 1e-30)
    if exclude is not None:
        keep = ~exclude.to(a.device).expand_as(a)
        excluded = 1.0 - float(keep.double().mean())
        a, b = a[keep], b[keep]
    else:
        excluded = 0.0
    a, b = a.flatten(), b.flatten()
    off = (a - b).abs() > tol * scale          # the few elements behind another discontinuity (sign(), == 0, floor)
    frac = float(off.double().mean())
    l2 = float((a - b)[~off].norm() / max(float(b.norm()), 1e-30))   # relative L2 error of all the others
    ok = l2 <= tol and frac <= max_outlier_frac and excluded <= 2e-2
    return ok, f"rel-L2 {l2:.2e}, outliers {frac:.2e}, max {float((a - b).abs().max()) / scale:.2e}, excluded {excluded:.2e}"


def min_select_ambiguous(tgt, src, K, depth_s, pose, trans_s, margin: float = 3e-5) -> torch.Tensor:
    """[B,H,W,1] bool: pixels whose gradient depends on a min-reprojection selection
    (``tf.minimum(e_0, e_1)``, SfMLearner_uni.py:168) that is decided by less than ``margin``
    relative - below the fp32 resolution of the SSIM term (sigma^2 = E[x^2] - E[x]^2 cancels
    ~3 digits), so two correct fp32 evaluation orders may legitimately pick different sides.
    The selection at q feeds the SSIM backward of the 3x3 window around q, hence the dilation.
    Evaluated with the stand-alone HIP operators (device tensors in, device mask out)."""
    import sfm_amd
    errs = []
    for i in range(2):
        g, _ = sfm_amd.projective_inverse_warp(src[..., 3 * i:3 * i + 3].contiguous(), depth_s, pose[:, i].contiguous(), K,
                                               add_trans=True, translations=trans_s[..., 3 * i:3 * i + 3].contiguous())
        errs.append(0.15 * (g - tgt).abs() + 0.85 * sfm_amd.tf_ssim(g, tgt, size=3, mean_metric=False))
    rel = ((errs[0] - errs[1]).abs() / torch.maximum(errs[0], errs[1]).clamp_min(1e-30)).amin(dim=3)
    near = ((rel > 0) & (rel < margin)).float()[:, None]  # exact ties (e.g. both warps off-image) are decided alike by every order
    return (torch.nn.functional.max_pool2d(near, 3, 1, 1)[:, 0] > 0)[..., None]


def write_snippet_dataset(root, n, H, W, n_src, seed=0, split_val=False):
    """A dataset directory in the reference's on-disk format (data/prepare_train_data.py:33-51, 99-113): per snippet a
    jpeg of the n_src+1 frames side by side and '<frame>_cam.txt' = 'fx,0.,cx,0.,fy,cy,0.,0.,1.', plus train.txt /
    val.txt listing 'folder frame'.  Smooth random content.  Returns (list of uint8 arrays as written, list of K)."""
    import os

    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    seqs, cams, lines = [], [], []
    for i in range(n):
        folder = "seq_%02d" % (i % 2)
        os.makedirs(os.path.join(root, folder), exist_ok=True)
        low = rng.uniform(0, 255, (H // 4 + 2, (n_src + 1) * W // 4 + 2, 3))
        arr = np.kron(low, np.ones((4, 4, 1)))[:H, :(n_src + 1) * W].astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(root, folder, "%06d.jpg" % i), quality=95)
        fx, fy, cx, cy = 241.7 + i, 246.3 + i, 204.2 - i, 59.0 + 0.5 * i
        with open(os.path.join(root, folder, "%06d_cam.txt" % i), "w") as f:
            f.write('%f,0.,%f,0.,%f,%f,0.,0.,1.' % (fx, cx, fy, cy))
        seqs.append(arr)
        cams.append(np.float32([[fx, 0, cx], [0, fy, cy], [0, 0, 1]]))
        lines.append("%s %06d\n" % (folder, i))
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.writelines(lines)
    with open(os.path.join(root, "val.txt"), "w") as f:
        f.writelines(lines[:max(1, n // 2)] if split_val else lines)
    return seqs, cams
