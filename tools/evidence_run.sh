#!/bin/bash
# Evidence run of a round (GPU box): the default bench line, its rocprofv3 kernel + memory-copy trace, the PMC passes.
# Usage: bash tools/evidence_run.sh <tag>   -> gpurun_out/<tag>/...   (summaries are then copied to profiles/<tag>_*)
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$2" != "nobench" ]; then timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1; fi
echo "bench done"
# the kernels at speed: the device-side leg alone (per-kernel averages that the bench line's HIP-event figures must match)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_device -o bench -- \
    python3 bench.py --seconds 4 --steps 0 --cpu-frames 0 --pipe-frames 0 > $out/prof_device_bench.json 2> $out/prof_device.err || exit 1
rm -f $out/prof_device/bench_kernel_trace.csv
# the export leg: kernels, blit copies (the device-to-host readback runs as __amd_rocclr_copyBuffer) and memory copies
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/prof_export -o bench -- \
    python3 bench.py --seconds 60 --steps 1 --warmup 1 --cpu-frames 0 --device-steps 0 --pipe-frames 0 > $out/prof_export_bench.json 2> $out/prof_export.err || exit 1
rm -f $out/prof_export/bench_kernel_trace.csv $out/prof_export/bench_memory_copy_trace.csv
echo "profile done"
bash tools/pmc_passes.sh $tag || exit 1
