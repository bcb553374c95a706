#!/bin/bash
# Round-2 evidence run (GPU box): the default bench line, its rocprofv3 kernel + memory-copy trace, the PMC passes.
# Usage: bash tools/r02_run.sh <tag>   -> gpurun_out/<tag>/...
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/prof -o bench -- \
    python3 bench.py --cpu-frames 0 > $out/prof_bench.json 2> $out/prof.err || exit 1
echo "profile done"
# the per-call traces are large: keep the statistics and the first 4000 lines of the copy trace
head -4000 $out/prof/bench_memory_copy_trace.csv > $out/prof/bench_memory_copy_trace_head.csv 2>/dev/null
rm -f $out/prof/bench_kernel_trace.csv $out/prof/bench_memory_copy_trace.csv
bash tools/pmc_passes.sh $tag || exit 1
