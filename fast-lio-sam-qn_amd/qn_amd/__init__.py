"""qn_amd - thin Python access to the MI355X-native registration engine (libqn_engine.so).

The product is the C-ABI library built from csrc/ (hand-written gfx950 HIP kernels) plus the
header-only C++ shims in shim/.  This package is plumbing for tests and bench.py: a ctypes
binding of include/qn_engine.h (engine.py), the build recipe (build.py) and the synthetic
scan-pair generator (synth.py).  There is NO CPU fallback anywhere in here.
"""
