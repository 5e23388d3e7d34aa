"""The allocation-free entry point sfm_splat_warp_fwd_static (include/sfm_amd.h; no counterpart in the reference: for
callers that cannot call hipMalloc) against the existing sfm_splat_warp_fwd, through the C ABI: two calls in a row, the
second with a smaller input (the library's device array keeps the first call's sums - the call must clear what it
uses), outputs within the path's 1e-5 (both use float atomics: the order of the adds differs run to run), masks
likewise; sizes beyond the array are rejected; the call allocates no device memory."""
import ctypes

import pytest
import torch

from sfm_amd import _lib, synth
from sfm_amd._lib import ptr, stream_of
from helpers import TOL_OUT, rel_err

pytestmark = pytest.mark.gpu


def _case(B, H, W, seed, dev):
    inp = synth.to_device(synth.make_inputs(B, H, W, S=1, N=2, seed=seed, pose_scale=0.02), dev)
    return (inp["tgt"], inp["depth"][0].squeeze(3).contiguous(), inp["trans"][0][..., :3].contiguous(),
            inp["pose"][:, 0].contiguous(), inp["K"])


def _both(lib, img, depth, trans, pose, K):
    B, H, W, C = img.shape
    st = stream_of(img)
    params = torch.empty(B, _lib.PARAM_STRIDE, device=img.device)
    assert lib.sfm_pose_setup(ptr(pose), 6, 0, ptr(K), B, 1, 0, ptr(params), st) == 0
    acc = torch.empty(B, H, W, C + 1, device=img.device)
    out_a, mask_a = torch.empty_like(img), torch.empty_like(img)
    out_b, mask_b = torch.full_like(img, 7.0), torch.full_like(img, 7.0)
    assert lib.sfm_splat_warp_fwd(ptr(img), ptr(depth), ptr(trans), ptr(params), B, H, W, C, ptr(acc), ptr(out_a), ptr(mask_a), st) == 0
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(img.device), torch.cuda.mem_get_info(img.device)[0]
    rc = lib.sfm_splat_warp_fwd_static(ptr(img), ptr(depth), ptr(trans), ptr(params), B, H, W, C, ptr(out_b), ptr(mask_b), st)
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated(img.device), torch.cuda.mem_get_info(img.device)[0]
    assert rc == 0
    return (out_a, mask_a), (out_b, mask_b), before, after


def test_static_entry_point_matches_the_workspace_one_twice_in_a_row(dev):
    lib = _lib.load()
    assert lib.sfm_splat_warp_fwd_static_max_floats() == 4 << 20
    with torch.cuda.device(dev):
        # first call also resolves the array's address on this device; the second is smaller and must not see the first's sums
        for k, (B, H, W, seed) in enumerate([(3, 96, 200, 91), (2, 37, 53, 92)]):
            (oa, ma), (ob, mb), before, after = _both(lib, *_case(B, H, W, seed, dev))
            assert rel_err(ob, oa) <= TOL_OUT, f"call {k}: out"
            assert rel_err(mb, ma) <= TOL_OUT, f"call {k}: mask"
            assert float(ob.abs().max()) > 0.1 and float(mb.max()) == 1.0
            if k == 1:  # steady state: no device memory taken by the call (torch's allocator untouched, free memory unchanged)
                assert after[0] == before[0] and after[1] == before[1]


def test_static_entry_point_rejects_sizes_beyond_its_array(dev):
    lib = _lib.load()
    with torch.cuda.device(dev):
        B, H, W = 1, 1024, 2048                     # 8 Mi floats of accumulator > 4 Mi
        img = torch.zeros(B, H, W, 3, device=dev)
        depth = torch.ones(B, H, W, device=dev)
        params = torch.zeros(B, _lib.PARAM_STRIDE, device=dev)
        out, mask = torch.empty_like(img), torch.empty_like(img)
        rc = lib.sfm_splat_warp_fwd_static(ptr(img), ptr(depth), None, ptr(params), B, H, W, 3, ptr(out), ptr(mask), stream_of(img))
        assert rc == -2                              # SFM_ERR_TOO_LARGE
        assert lib.sfm_splat_warp_fwd_static(None, ptr(depth), None, ptr(params), 1, 8, 8, 3, ptr(out), ptr(mask), stream_of(img)) == -1
