"""CPU checks of the C-ABI boundary: the library builds for gfx950 without a
GPU, loads, and exports exactly what include/pmf_hip.h declares; the product
fails loudly without the library and never touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "prob-matrix-factorization_amd")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import pmf_hip
    return pmf_hip


def _declared():
    text = open(os.path.join(ROOT, "include", "pmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(pmf_[a-z0-9_]+)\s*\(", text))


def test_header_and_binding_list_the_same_functions(built):
    assert _declared() == set(built.SIGNATURES)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert built.load().pmf_abi_version() == 1


def test_no_gpu_is_an_error_not_a_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert built.device_count() == 0
    with pytest.raises(built.PmfError):
        built.Context(10, 10, 8)
    msg = built.load().pmf_last_error().decode()
    assert "hip" in msg.lower() or "device" in msg.lower()


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", os.path.join(PKG, "pmf_hip", "does_not_exist.so"))
    with pytest.raises(built.PmfLibraryError, match="no CPU fallback"):
        built.load()


def test_model_fit_without_gpu_raises(built):
    import numpy as np
    import pandas as pd
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
    df = pd.DataFrame({"u": [0, 1, 2], "i": [0, 1, 1], "rating": [1.0, 2.0, 3.0]})
    with pytest.raises(RuntimeError):
        HPF_CAVI(HPF_CAVI_Config(n_factors=4, max_iter=1, verbose=False)).fit(df)


def test_product_never_imports_the_oracle():
    offenders = []
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "cavi_oracle" in text:
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_kernels_target_gfx950_only(built):
    """Every embedded code object is gfx950 (no multi-arch / fallback bundles)."""
    blob = open(built.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets
