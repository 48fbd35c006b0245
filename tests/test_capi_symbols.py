"""CPU-side boundary checks: the C-ABI library loads, exports every symbol include/qn_engine.h
declares, and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "qn_engine.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qn_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_reference_surface():
    syms = declared_symbols()
    for s in ["qn_ctx_create", "qn_gicp_set_source", "qn_gicp_set_target", "qn_gicp_compute_covariances",
              "qn_gicp_align", "qn_gicp_fitness", "qn_gicp_transformed_source", "qn_icp_alignment",
              "qn_quatro_set_params", "qn_quatro_align", "qn_fpfh", "qn_match_optimized", "qn_coarse_to_fine_alignment", "qn_icp_alignment_batch"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    from qn_amd import build, engine
    build.build()
    lib = ctypes.CDLL(build.LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qn_amd import engine
    with pytest.raises(engine.EngineError) as ei:
        engine.Context(1000)
    assert ei.value.status == engine.QN_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """The product path must not route through oracle/ (only tests/, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "fast-lio-sam-qn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle", src, flags=re.M), f
