#!/bin/bash
# Scheduler-strategy variants of stft.o (same source; only -mllvm flags differ), timed with tools/stft_bench.py.
# Build (CPU): see the loop in profiles/r01_stft_sched.txt; run on the GPU box: bash tools/stft_sched.sh
cd "$(dirname "$0")/.."
for v in 0 1 2 3 4 5 6; do
  echo "== v$v"
  AURORA_HIP_LIB=$PWD/build_sched/lib_v$v.so timeout -k 10 100 python tools/stft_bench.py --reps 10 2>&1 | grep -E "stft:" || exit 1
done
