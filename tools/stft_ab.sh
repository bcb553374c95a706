#!/bin/bash
# Interleaved A/B timing of STFT kernel variants built under build_lesion/ (tools/stft_lesion.sh or by hand):
#   VARIANTS="old new" ROUNDS=3 bash tools/stft_ab.sh      (GPU box; AUR_STFT_FPW etc. pass through)
cd "$(dirname "$0")/.."
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in ${VARIANTS:-old new}; do
    echo -n "round $r $v: "
    AURORA_HIP_LIB=$PWD/build_lesion/lib_$v.so timeout -k 10 100 python tools/stft_bench.py --reps ${REPS:-10} 2>&1 | grep stft: || exit 1
  done
done
