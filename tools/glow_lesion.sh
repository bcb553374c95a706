#!/bin/bash
# Where the glow kernel's time goes (GPU box): shipped library vs builds with the layer loop removed (GLOW_LESION=1) and
# with one segment per group scanned instead of all (GLOW_LESION=2), ending after the region test (3), empty (4).
#   CPU:  bash tools/glow_lesion.sh build     (two extra libraries under build_glow/, which travel with gpurun)
#   GPU:  bash tools/glow_lesion.sh
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    for v in 1 2 3 4; do
        make -s -C aurora_amd/csrc -j8 BUILD=/tmp/glow_lesion_b$v OUT=$PWD/build_glow/lib_l$v.so EXTRA=-DGLOW_LESION=$v || exit 1
    done
    exit 0
fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in shipped l1 l2 l3 l4; do
    echo "== $lib"
    if [ $lib = shipped ]; then unset AURORA_HIP_LIB; else export AURORA_HIP_LIB=$PWD/build_glow/lib_$lib.so; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/glow_lesion_$lib -o g -- \
        python3 bench.py --mode "Spectrum - Radial" --glow --steps 0 --cpu-frames 0 --device-seconds 60 --device-steps 2 \
        > gpurun_out/glow_lesion_$lib.json 2> gpurun_out/glow_lesion_$lib.err || { tail -5 gpurun_out/glow_lesion_$lib.err; exit 1; }
    rm -f gpurun_out/glow_lesion_$lib/g_kernel_trace.csv
    grep -h "glow_kernel" gpurun_out/glow_lesion_$lib/g_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
