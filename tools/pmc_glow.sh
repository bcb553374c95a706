#!/bin/bash
# SQ counters of the glow kernel (GPU box): bash tools/pmc_glow.sh  -> gpurun_out/pmc_glow/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_glow
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY \
    --output-format csv -d $out/sq -o p -- python3 bench.py --mode "Spectrum - Radial" --glow --steps 0 --cpu-frames 0 \
    --device-seconds 8 --device-steps 1 --device-warmup 0 > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
python3 - <<'PY'
import csv, collections, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/pmc_glow/sq/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "glow_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_glow/summary.txt", "w") as out:
    for name, d in sorted(rows.items()):
        vals = [sum(v) for v in d.values()]
        line = f"{name:24s} mean per dispatch {sum(vals) / len(vals):16.0f}   dispatches {len(vals)}"
        print(line)
        out.write(line + "\n")
PY
rm -f $out/sq/*counter_collection.csv
