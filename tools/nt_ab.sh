#!/bin/bash
# A/B of non-temporal frame stores in the fill kernels (GPU box): shipped library vs a -DAUR_STREAM_STORES=0 build
# (make -C aurora_amd/csrc BUILD=/tmp/b_plain OUT=../../build_nt/lib_plain.so EXTRA=-DAUR_STREAM_STORES=0).
cd "$(dirname "$0")/.."
show() { tail -1 "$1" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("  device frames/s %.0f  %s %.1f GB/s  %.4f ms/launch" % (d["device_fps"], r["kernel"], r["achieved"], r["avg_launch_ms"]))'; }
for mode in "Spectrum - Linear" "Spectrum - Radial" "Waveform - Linear"; do
  for lib in shipped plain; do
    echo "== $mode / $lib"
    if [ $lib = plain ]; then export AURORA_HIP_LIB=$PWD/build_nt/lib_plain.so; else unset AURORA_HIP_LIB; fi
    timeout -k 10 150 python -u bench.py --mode "$mode" --steps 0 --cpu-frames 0 --device-seconds 600 --device-steps 2 > gpurun_out/nt_ab.tmp 2>&1 || { tail -5 gpurun_out/nt_ab.tmp; exit 1; }
    show gpurun_out/nt_ab.tmp
  done
done
