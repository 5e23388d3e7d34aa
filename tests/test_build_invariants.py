"""Build invariants of the hand-scheduled kernels, checked on the device assembly hipcc emits (no GPU needed):
  * no kernel of the fused loss keeps state in scratch memory (private_segment_fixed_size == 0, no spills) - a `?:`
    between struct members or a float4 member once put a whole wave's state there, 7x slower with every test green;
  * the streaming stencil has no VALU instruction that writes EXEC (v_cmpx, v_readlane ...): its DPP sums are inline asm,
    which the compiler's hazard recogniser does not see, and "VALU writes EXEC -> DPP" needs 5 wait states where the
    blocks only guarantee the 2 of "VALU writes VGPR -> DPP reads it"."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sfm_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "--cuda-device-only", "-S"]


def _asm(src, contract, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    subprocess.check_call([HIPCC, *FLAGS, f"-ffp-contract={contract}", os.path.join(CSRC, src), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text):
    names = re.findall(r"\.name:\s+(\S+)", text)
    priv = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
    spill = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    vgpr = [int(v) for v in re.findall(r"\.vgpr_count:\s+(\d+)", text)]
    assert len(names) == len(priv) == len(spill) == len(vgpr) and names
    return list(zip(names, priv, spill, vgpr))


def test_stencil_kernels_have_no_scratch_and_no_exec_writing_valu(tmp_path):
    text = _asm("loss_stencil.hip", "fast", tmp_path)
    ks = _kernels(text)
    # gather, splat (each with and without seeds), the seed-guard (redo) form of the splat side, and the reduction of the
    # per-wave sums with and without the guard inlined
    assert len(ks) == 7
    for name, priv, spill, vgpr in ks:
        assert priv == 0 and spill == 0, f"{name}: {priv} bytes of scratch, {spill} spilled VGPRs"
    assert "v_fmac_f32_dpp" in text, "the hand-written DPP sums are gone"
    # per kernel: nothing of that kind from the first DPP instruction on (scalar index arithmetic in a kernel's prologue -
    # uniform integer divisions end in v_readfirstlane - is far from the row loop and harmless)
    bodies = [b for b in re.split(r"\n\s*\.section\s+", text)[1:] if b.startswith(".text.") and "s_endpgm" in b]
    assert len(bodies) == len(ks)
    for body in bodies:
        first = re.search(r"\bv_\w+_dpp\b", body)
        if "loss_reduce_kernelILb0" in body[:200]:
            assert not first  # the plain reduction has no stencil in it
            continue
        assert first, "a stencil kernel without DPP sums"
        assert not re.search(r"\bv_cmpx|\bv_readlane|\bv_readfirstlane|\bv_writelane", body[first.start():]), \
            "a VALU write of EXEC / SGPR next to inline-asm DPP"
    occ = {n: 512 // v for n, _, _, v in ks}
    assert all(o >= 2 for o in occ.values()), occ          # waves per SIMD the register budget allows
    assert min(o for n, o in occ.items() if "splat" in n and "ILb1" in n) >= 3, occ


def test_loss_kernels_have_no_scratch(tmp_path):
    ks = _kernels(_asm("loss.hip", "off", tmp_path))
    for name, priv, spill, vgpr in ks:
        assert priv == 0 and spill == 0, f"{name}: {priv} bytes of scratch, {spill} spilled VGPRs"
    bwd = [v for n, _, _, v in ks if "loss_warp_bwd_kernel" in n]
    assert bwd and max(bwd) <= 168, f"backward warp above the three-waves-per-SIMD budget: {bwd}"
