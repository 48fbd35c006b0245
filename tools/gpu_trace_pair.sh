#!/bin/bash
# rocprofv3 kernel trace of single-stream registrations of chosen synthetic pairs: usage tools/gpu_trace_pair.sh <tag> '<json knobs>' <pair ids...>
TAG=${1:-trp}; KN="${2:-{\}}"; shift; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
QN_DEBUG_KNOBS="$KN" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python tools/gpu_probe_pairs.py $@ > $OUT/probe.txt 2> $OUT/prof.err; echo "rocprof exit $?"
find $OUT/prof -name '*kernel_trace.csv' -exec sh -c 'python tools/trace_summary.py {} > '$OUT'/last_registration_trace.txt 2>&1' \;
rm -rf $OUT/prof
cat $OUT/last_registration_trace.txt | cut -c1-150
