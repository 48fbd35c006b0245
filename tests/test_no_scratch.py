"""Every kernel of the default registration chain (k <= 24 neighbours) must run WITHOUT scratch: a dispatch whose code object has a private segment pays at the kernel
boundary on gfx950 (tools/micro/launch_gap2.hip; DESIGN.md 4c "No scratch": k_tick 68 -> 0 B was +1.6 %, the list kernels +2.3 % of the headline).  Reads the metadata of the
built library with tools/scratch_report.py - no GPU needed.  Allowed to keep scratch: the k = 25..32 sorted-list tail (KnnCovK<32>) and the developer clock-probe ticks."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALLOWED = (r"KnnCovK<32,", r"k_knn_cov<32,", r"k_tick<\d+, \d+, \d+, true>", r"k_align_persist<\d+, true>")


def test_default_chain_kernels_have_no_scratch():
    lib = os.path.join(ROOT, "fast-lio-sam-qn_amd", "libqn_engine.so")
    if not os.path.exists(lib):
        from qn_amd import build
        build.build(verbose=False)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scratch_report.py")], capture_output=True, text=True, check=True).stdout
    rows = [l for l in out.splitlines() if " B scratch" in l]
    assert out.strip().splitlines()[-1].split()[0].isdigit(), out[-300:]          # "<n> kernels, <m> with scratch"
    bad = [l.strip() for l in rows if int(l.split()[0]) > 0 and not any(re.search(a, l) for a in ALLOWED)]
    assert not bad, "kernels of the chain with scratch:\n" + "\n".join(bad)
