"""csrc/qn_pool.h - the parked host workers behind qn_icp_alignment_batch / qn_coarse_to_fine_align_batch / qn_multi_align_best (one worker per context = stream, one per
GPU; the reference's own fan-out point is the single timer thread of fast_lio_sam_qn.cpp:213-219).  Host-only C++: compiled with g++ and run here, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_worker_pool_fan_out_nesting_and_reuse(tmp_path):
    exe = str(tmp_path / "pool_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "fast-lio-sam-qn_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "pool_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
