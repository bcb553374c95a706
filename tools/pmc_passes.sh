#!/bin/bash
# rocprofv3 counter passes for bench.py (GPU box).  One --pmc pass per run (no trace domains combined).
# Usage: bash tools/pmc_passes.sh <tag>      -> gpurun_out/pmc_<tag>/{fetch,write,sq}/..._counter_collection.csv
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the device-side leg alone (no export leg: counter collection serialises the queues, the readback pipeline is left
# out): one warm-up and one timed step over the 1 h unit
eval "EXTRA=(${BENCH_EXTRA:-})"     # e.g. BENCH_EXTRA='--mode "Spectrum - Radial" --device-seconds 120'
# rocprofv3's counter mode faults in its own dispatch intercept for some dispatch sequences of this program
# (profiles/r03_pmc_fault_probe.txt: HIP events, the instrumented kernels and the dispatch count as such were ruled out;
# it is where a launch's data sits when the intercept copies it).  This shape -- no warm-up step, 128-frame batches, every
# kernel instrumented -- ran all passes in rounds 2 and 3; the fill's traffic is linear in the frames of a launch.
# PMC_KERNELS (a regex) limits the instrumented kernels.  A pass that fails ends the script: nothing runs after a fault.
FILTER=()
if [ -n "$PMC_KERNELS" ]; then FILTER=(--kernel-include-regex "$PMC_KERNELS"); fi
export AUR_PMC_FILL_FRAMES=${AUR_PMC_FILL_FRAMES:-128}
BENCH=(bench.py --seconds 4 --steps 0 --cpu-frames 0 --pipe-frames 0 --device-steps 1 --device-warmup 0 --batch 128 "${EXTRA[@]}")
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE "${FILTER[@]}" --output-format csv -d $out/fetch -o p -- python3 "${BENCH[@]}" > $out/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE "${FILTER[@]}" --output-format csv -d $out/write -o p -- python3 "${BENCH[@]}" > $out/write.log 2>&1
echo "write pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY "${FILTER[@]}" --output-format csv -d $out/sq -o p -- python3 "${BENCH[@]}" > $out/sq.log 2>&1
echo "sq pass done"
python3 tools/pmc_summarize.py $out > $out/summary.txt
cat $out/summary.txt
# keep only the aggregated rows: the raw per-dispatch CSVs are large
find $out -name "*_counter_collection.csv" -size +8M -delete
