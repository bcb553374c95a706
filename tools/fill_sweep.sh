#!/bin/bash
# Sweep of the fill kernel's tuning knobs (GPU box). Usage: bash tools_fill_sweep.sh
set -e
for nt in 0 1; do for rows in 8 16 24 48 96; do
  echo "nt=$nt rows=$rows"
  AUR_FILL_NT=$nt AUR_FILL_ROWS=$rows timeout -k 10 120 python -u bench.py --seconds 600 --steps 2 --warmup 1 --cpu-frames 0 --pcie-frames 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'])"
done; done
