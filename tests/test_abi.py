"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly the
symbols include/sfm_amd.h declares (no compute calls without a GPU), and the Python
mirror rejects what it must."""
import os
import re

import pytest
import torch

from sfm_amd import _lib
import sfm_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sfm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sfm_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_arity_matches_header():
    text = open(os.path.join(ROOT, "include", "sfm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        assert len([a for a in args.split(",") if a.strip() and a.strip() != "void"]) == len(_lib.SIGNATURES[name]), name


def test_pure_host_entry_points():
    lib = _lib.load()
    assert lib.sfm_bwd_blocks(128, 416) == (128 * 416 + 1023) // 1024
    assert lib.sfm_loss_num_partials(8, 128, 416, 4, 2) == 4 * 7
    assert lib.sfm_loss_workspace_floats(8, 128, 416, 4, 2) > 8 * 128 * 416 * 4 * 2 * 8
    assert lib.sfm_loss_workspace_floats(8, 128, 416, 4, 1) == -1  # needs two sources (SfMLearner_uni.py:168)
    assert lib.sfm_loss_workspace_floats(8, 128, 416, 4, 5) == -1


def test_cpu_tensors_are_rejected_not_emulated():
    x = torch.zeros(1, 4, 4, 3)
    with pytest.raises(TypeError):
        sfm_amd.projective_inverse_warp(x, torch.ones(1, 4, 4), torch.zeros(1, 6), torch.eye(3)[None])
    with pytest.raises(TypeError):
        sfm_amd.bilinear_sampler(x, torch.zeros(1, 4, 4, 2))
    with pytest.raises(TypeError):
        sfm_amd.tf_ssim(x.double(), x.double())


def test_no_oracle_in_product_path():
    """the package must never import the oracle"""
    pkg = os.path.join(ROOT, "sfm_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+[\w.]*oracle", src, flags=re.M), f
