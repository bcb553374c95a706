#!/bin/bash
# Round 2 saw `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --seconds 4 --steps 0 --cpu-frames 0 --device-steps 1` die with
# SIGSEGV right after HSA init (gpurun_out/pmc_r02a, pmc_r02b), while the same pass with a quarter of the dispatches ran.
# Two passes at the ORIGINAL dispatch count, one run each, to tell the causes apart:
#   A: no HIP events recorded by the bench (--events 0)
#   B: events as before, counters collected for the two kernels of interest only (--kernel-include-regex)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_fault_r03
mkdir -p $out
BENCH=(bench.py --seconds 4 --steps 0 --cpu-frames 0 --device-steps 1 --pipe-frames 0)
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/a -o p -- python3 "${BENCH[@]}" --events 0 > $out/a.log 2>&1
echo "A (no events) exit $?" | tee -a $out/verdict.txt
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "raster_fill_lin_kernel|stft_frames_kernel" --output-format csv -d $out/b -o p -- python3 "${BENCH[@]}" > $out/b.log 2>&1
echo "B (two kernels only) exit $?" | tee -a $out/verdict.txt
for k in a b; do
  f=$(find $out/$k -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && echo "$k: $(wc -l < $f) counter rows" | tee -a $out/verdict.txt
  find $out/$k -name "*_counter_collection.csv" -size +8M -delete
done
tail -3 $out/a.log | cut -c1-300 > $out/a.tail; tail -3 $out/b.log | cut -c1-300 > $out/b.tail
exit 0
