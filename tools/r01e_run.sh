#!/bin/bash
# Round-1 final profiles after the fill kernels moved to non-temporal stores (GPU box): rocprofv3 kernel stats of the
# default bench and of "Spectrum - Radial", then the PMC traffic passes.   Usage: bash tools/r01e_run.sh
set -e
mkdir -p gpurun_out/r01e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01e/bench -o p -- python3 bench.py --cpu-frames 0 --pcie-frames 0 > gpurun_out/r01e/bench.log 2>&1
echo "bench ok"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01e/radial -o p -- python3 bench.py --mode "Spectrum - Radial" --seconds 240 --cpu-frames 0 --pcie-frames 0 > gpurun_out/r01e/radial.log 2>&1
echo "radial ok"
find gpurun_out/r01e -name "*kernel_trace.csv" -delete
bash tools/pmc_passes.sh r01e
