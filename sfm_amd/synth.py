"""Seeded synthetic KITTI-shaped inputs for tests and bench.py.

Generated on the CPU with a fixed ``torch.Generator`` so that the oracle and the HIP
path see bit-identical tensors wherever they run; callers move them to the device.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

SEED = 8964  # the reference's seed (train_uni.py:32)

# KITTI raw intrinsics at 1242x375 (data/kitti/kitti_raw_loader.py:96-107,146-152 scales them)
_KITTI = dict(fx=721.5377, fy=721.5377, cx=609.5593, cy=172.854, w=1242.0, h=375.0)
_CITY = dict(fx=2262.52, fy=2265.3017, cx=1096.98, cy=513.137, w=2048.0, h=1024.0)


def intrinsics(B: int, H: int, W: int, cityscapes: bool = False) -> torch.Tensor:
    k = _CITY if cityscapes else _KITTI
    sx, sy = W / k["w"], H / k["h"]
    m = torch.tensor([[k["fx"] * sx, 0.0, k["cx"] * sx], [0.0, k["fy"] * sy, k["cy"] * sy], [0.0, 0.0, 1.0]],
                     dtype=torch.float32)
    return m[None].repeat(B, 1, 1).contiguous()


def _smooth(g: torch.Generator, B: int, H: int, W: int, C: int, k: int = 4) -> torch.Tensor:
    """Low-pass filtered noise in NHWC: a coarse random field upsampled bilinearly."""
    h, w = max(H // k, 2), max(W // k, 2)
    z = torch.rand(B, C, h, w, generator=g) * 2 - 1
    z = F.interpolate(z, size=(H, W), mode="bilinear", align_corners=True)
    return z.permute(0, 2, 3, 1).contiguous()


def make_inputs(B: int, H: int, W: int, S: int = 4, N: int = 2, seed: int = SEED, pose_scale: float = 0.01,
                full_res: bool = True, cityscapes: bool = False) -> Dict[str, object]:
    """tgt [B,H,W,3], src [B,H,W,3N] in [-1,1]; K [B,3,3]; pose [B,N,6] ~ pose_scale*N(0,1);
    depth[s] [B,h,w,1] = 1/disp with disp = 10*sigmoid(n)+0.01 normalised to mean 1 per
    item (nets.py:11-12,689; SfMLearner_uni.py:40,47); trans[s] [B,h,w,3N] ~ 0.01*smooth
    (nn_nets.py:120-134).  full_res=True gives every scale at HxW (what the warp sees after
    the resize of SfMLearner_uni.py:91,113); else scale s is at H/2^s x W/2^s."""
    g = torch.Generator().manual_seed(seed)
    tgt = _smooth(g, B, H, W, 3)
    src = torch.clamp(_smooth(g, B, H, W, 3 * N) * 0.5 + tgt.repeat(1, 1, 1, N) * 0.5, -1, 1).contiguous()
    pose = (torch.randn(B, N, 6, generator=g) * pose_scale).contiguous()
    depth: List[torch.Tensor] = []
    trans: List[torch.Tensor] = []
    disp: List[torch.Tensor] = []
    for s in range(S):
        h, w = (H, W) if full_res else (max(H >> s, 1), max(W >> s, 1))
        n = _smooth(g, B, h, w, 1, k=8) * 2.0
        d = 10.0 * torch.sigmoid(n) + 0.01
        d = d / d.mean(dim=(1, 2, 3), keepdim=True)
        disp.append(d.contiguous())
        depth.append((1.0 / d).contiguous())
        trans.append((0.01 * _smooth(g, B, h, w, 3 * N, k=8)).contiguous())
    return dict(tgt=tgt, src=src, K=intrinsics(B, H, W, cityscapes), pose=pose, depth=depth, trans=trans, disp=disp)


def to_device(inp: Dict[str, object], device) -> Dict[str, object]:
    out = {}
    for k, v in inp.items():
        out[k] = [t.to(device) for t in v] if isinstance(v, list) else v.to(device)
    return out


def make_step_inputs_device(B: int, H: int, W: int, S: int, N: int, seed: int, device, pose_scale: float = 0.01,
                            cityscapes: bool = False, relative_field: bool = False) -> Dict[str, torch.Tensor]:
    """The tensors one fused step reads, generated ON the device (same distributions as ``make_inputs`` with
    ``full_res=True``, a device generator - not bit-identical to the CPU stream): for the large benchmark
    configurations, where CPU generation and the host -> device copy of tens of GB would dominate the run.
    Returns tgt [B,H,W,3], src [B,H,W,3N], K [B,3,3], pose [B,N,6], depth [S,B,H,W], trans [S,B,H,W,3N] (stacked,
    the layout sfm_loss_forward reads).  ``relative_field``: the depth / translation noise lives on the 16 x 52 grid
    the 128 x 416 configuration has (one cell = 1/52 of the image width at every resolution) instead of one cell per
    8 pixels - the same scene content at a higher resolution, rather than 4x finer depth detail."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)

    def smooth(b, c, k):
        h, w = max(H // k, 2), max(W // k, 2)
        if relative_field and k == 8:
            h, w = 16, 52
        z = torch.rand(b, c, h, w, generator=g, device=dev) * 2 - 1
        z = F.interpolate(z, size=(H, W), mode="bilinear", align_corners=True)
        return z.permute(0, 2, 3, 1)

    tgt = smooth(B, 3, 4).contiguous()
    src = torch.empty(B, H, W, 3 * N, device=dev)
    for i in range(N):
        src[..., 3 * i:3 * i + 3] = torch.clamp(smooth(B, 3, 4) * 0.5 + tgt * 0.5, -1, 1)
    pose = (torch.randn(B, N, 6, generator=g, device=dev) * pose_scale).contiguous()
    depth = torch.empty(S, B, H, W, device=dev)
    trans = torch.empty(S, B, H, W, 3 * N, device=dev)
    for s in range(S):
        d = 10.0 * torch.sigmoid(smooth(B, 1, 8) * 2.0) + 0.01
        d = d / d.mean(dim=(1, 2, 3), keepdim=True)
        depth[s] = (1.0 / d).squeeze(3)
        for i in range(N):  # per source: the temporaries stay small at 2048x1024
            trans[s, ..., 3 * i:3 * i + 3] = 0.01 * smooth(B, 3, 8)
    return dict(tgt=tgt, src=src, K=intrinsics(B, H, W, cityscapes).to(dev), pose=pose, depth=depth, trans=trans)
