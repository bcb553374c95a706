#!/bin/bash
# Round-2 evidence for the closed-path modes at 4K: device-side rate and per-kernel times of "Spectrum - Radial" with and
# without the glow decoration (gpurun: bash tools/r02_radial.sh <tag>).
tag=${1:-r02_radial}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for variant in plain glow; do
    extra=""; [ $variant = glow ] && extra="--glow"
    timeout -k 10 300 python3 bench.py --mode "Spectrum - Radial" $extra --seconds 60 --steps 0 --cpu-frames 0 \
        --device-seconds 240 --device-steps 2 > $out/bench_$variant.json 2> $out/bench_$variant.err || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$variant -o radial -- \
        python3 bench.py --mode "Spectrum - Radial" $extra --seconds 60 --steps 0 --cpu-frames 0 --device-seconds 60 \
        --device-steps 2 > $out/prof_$variant.json 2> $out/prof_$variant.err || exit 1
    rm -f $out/prof_$variant/radial_kernel_trace.csv
done
echo done
