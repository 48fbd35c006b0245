"""The header-only drop-in shim (nano_gicp/nano_gicp.hpp) compiles against stand-in pcl/Eigen headers,
links against the C-ABI library and - on a GPU - reproduces LoopClosure::icpAlignment
(fast_lio_sam_qn/src/loop_closure.cpp:110-136) with the oracle's answer."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "shim_icp_alignment")


def build_shim_program():
    from qn_amd import build
    build.build()
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "tests", "standins"),
           "-I" + os.path.join(ROOT, "fast-lio-sam-qn_amd", "shim"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "shim_icp_alignment.cpp"), "-L" + os.path.join(ROOT, "fast-lio-sam-qn_amd"),
           "-lqn_engine", "-Wl,-rpath," + os.path.join(ROOT, "fast-lio-sam-qn_amd"), "-o", BIN]
    subprocess.check_call(cmd)
    return BIN


def test_shim_compiles_and_links():
    assert os.path.exists(build_shim_program())


@pytest.mark.gpu
def test_shim_reproduces_icp_alignment(tmp_path, oracle):
    from qn_amd import synth
    if not os.path.exists(BIN):
        build_shim_program()
    src, tgt, T = synth.make_pair(70, 5000, extent=45.0)
    a, b = tmp_path / "src.bin", tmp_path / "dst.bin"
    src.tofile(a); tgt.tofile(b)
    out = subprocess.check_output([BIN, str(a), str(b)]).decode().split()
    valid, conv, score = int(out[0]), int(out[1]), float(out[2])
    Tm = np.array([float(x) for x in out[3:19]]).reshape(4, 4)
    ro = oracle.icp_alignment(src, tgt)
    assert bool(valid) == ro["valid"] and bool(conv) == ro["converged"]
    assert abs(score - ro["score"]) <= 1e-6 * ro["score"]
    dt, dr = synth.pose_error(Tm, ro["T"])
    assert dt <= 1e-4 and dr <= 1e-4
    n_out, x0, inten = int(out[19]), float(out[20]), float(out[21])
    assert n_out == len(src) and inten == 42.0                     # output cloud: source fields kept, xyz transformed
    assert abs(x0 - ro["raw"]["Tf"][0] @ np.r_[src[0], 1.0]) < 1e-3
