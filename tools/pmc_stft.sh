#!/bin/bash
# SQ counter passes for the STFT kernel alone (GPU box): bash tools/pmc_stft.sh <tag>
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcstft_$tag
mkdir -p $out
CMD="tools/stft_bench.py --reps 2"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/a -o p -- python3 $CMD > $out/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/b -o p -- python3 $CMD > $out/b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/c -o p -- python3 $CMD > $out/c.log 2>&1 || echo "pass c failed"
python3 - $out <<'PY'
import csv,glob,sys,os
from collections import defaultdict
agg=defaultdict(lambda:[0.0,0])
for p in glob.glob(os.path.join(sys.argv[1],'*','**','*counter_collection.csv'),recursive=True):
    for r in csv.DictReader(open(p)):
        if 'stft_frames_kernel' in r['Kernel_Name']:
            a=agg[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,v in sorted(agg.items()): print(f"{k:28s} {v[0]/v[1]:16.1f}  (n={v[1]})")
PY
find $out -name "*_counter_collection.csv" -delete
