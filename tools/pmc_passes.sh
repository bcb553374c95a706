#!/bin/bash
# rocprofv3 counter passes for bench.py (GPU box).  One --pmc pass per run (no trace domains combined).
# Usage: bash tools/pmc_passes.sh <tag>      -> gpurun_out/pmc_<tag>/{fetch,write,sq}/..._counter_collection.csv
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the device-side leg alone (no export leg: counter collection serialises the queues, the readback pipeline is left
# out): one warm-up and one timed step over the 1 h unit
eval "EXTRA=(${BENCH_EXTRA:-})"     # e.g. BENCH_EXTRA='--mode "Spectrum - Radial" --device-seconds 120'
# Counters are collected for the kernels named in KERNELS only: rocprofv3's counter mode faults in its own dispatch
# intercept once about 10 000 dispatches of one process are instrumented (profiles/r03_pmc_fault_probe.txt: HIP events and
# the program's own dispatch count were ruled out; 6 753 instrumented dispatches run, 10 127 fault), and the device leg
# makes three launches per batch.  With the filter the leg runs with its real launches (default batch).
KERNELS=${PMC_KERNELS:-"raster_fill_lin_kernel|stft_frames_kernel|raster_fill_rows_kernel|paint_stroke_closed_kernel|scan_frames_kernel"}
BENCH=(bench.py --seconds 4 --steps 0 --cpu-frames 0 --pipe-frames 0 --device-steps 1 --device-warmup 0 "${EXTRA[@]}")
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $out/fetch -o p -- python3 "${BENCH[@]}" > $out/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $out/write -o p -- python3 "${BENCH[@]}" > $out/write.log 2>&1
echo "write pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-include-regex "$KERNELS" --output-format csv -d $out/sq -o p -- python3 "${BENCH[@]}" > $out/sq.log 2>&1
echo "sq pass done"
python3 tools/pmc_summarize.py $out > $out/summary.txt
cat $out/summary.txt
# keep only the aggregated rows: the raw per-dispatch CSVs are large
find $out -name "*_counter_collection.csv" -size +8M -delete
