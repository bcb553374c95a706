#!/bin/bash
# A/B of the lanes that share one stroke primitive in paint_stroke_closed_kernel (GPU box; libraries built on the CPU with
#   make -C aurora_amd/csrc BUILD=/tmp/pl_b$v OUT=$PWD/build_pl/lib_$v.so EXTRA=-DAUR_PAINT_LANES=$v   for v in 4 16 32).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in ${LANES:-8 4 16 32 8}; do
    if [ $v = 8 ]; then unset AURORA_HIP_LIB; else export AURORA_HIP_LIB=$PWD/build_pl/lib_$v.so; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pl_$v -o g -- \
        python3 bench.py --mode "${MODE:-Spectrum - Radial}" --steps 0 --cpu-frames 0 --device-seconds 120 --device-steps 2 \
        > gpurun_out/pl_$v.json 2> gpurun_out/pl_$v.err || { tail -5 gpurun_out/pl_$v.err; exit 1; }
    rm -f gpurun_out/pl_$v/g_kernel_trace.csv
    echo "lanes $v: $(grep -h paint_stroke gpurun_out/pl_$v/g_kernel_stats.csv | cut -d, -f4) ns stroke; $(python3 -c "import json; d=json.loads(open('gpurun_out/pl_$v.json').read().strip().splitlines()[-1]); print(d['device_fps'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"
done
