#!/bin/bash
# Round-1 late profiles (GPU box): decoration tests, aur_add_planes alone, then rocprofv3 kernel stats of the configs[3]
# union workload and of "Spectrum - Radial" with the glow decoration.   Usage: bash tools/r01d_run.sh
set -e
mkdir -p gpurun_out/r01d
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "glow or add_planes or worker or shadow or fill or feather or blend or composite" > gpurun_out/r01d/glow_tests.log 2>&1
echo "tests ok"
timeout -k 10 100 python -u tools/add_bench.py > gpurun_out/r01d/add_bench.txt 2>&1
echo "add_bench ok"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d/cfg4 -o p -- python3 bench.py --mode "Particles + Feather" --width 1920 --height 1080 --seconds 240 --cpu-frames 0 --pcie-frames 0 > gpurun_out/r01d/cfg4.log 2>&1
echo "cfg4 ok"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d/glow -o p -- python3 bench.py --mode "Spectrum - Radial" --glow --seconds 240 --cpu-frames 0 --pcie-frames 0 > gpurun_out/r01d/glow.log 2>&1
echo "glow ok"
find gpurun_out/r01d -name "*kernel_trace.csv" -delete
tail -8 gpurun_out/r01d/add_bench.txt
