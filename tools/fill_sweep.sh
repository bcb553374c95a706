#!/bin/bash
# Sweep of the fill kernel's tuning knobs (GPU box). Usage: bash tools/fill_sweep.sh (device-side leg of bench.py)
set -e
for nt in 0 1; do for rows in 8 16 24 48 96; do
  echo "nt=$nt rows=$rows"
  AUR_FILL_NT=$nt AUR_FILL_ROWS=$rows timeout -k 10 120 python -u bench.py --steps 0 --cpu-frames 0 --device-seconds 600 --device-steps 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['device_fps'], d['roofline']['achieved'])"
done; done
