#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B?  No: on gfx950 they are in KB
(kilobytes); MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request on wide streaming reads,
so it is doubled before it is compared with a byte count; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("stft_frames_kernel", "raster_fill_lin_kernel", "raster_fill16_kernel", "raster_fill_kernel",
                "span_summary_kernel", "scan_frames_kernel", "spans_spectrum_kernel", "spans_waveform_kernel",
                "spans_particles_kernel", "raster_fill_rows_kernel", "paint_stroke_closed_kernel",
                "raster_patch_planes_kernel", "glow_kernel"):
        if key in name:
            return key
    return None


def main(root):
    # a kernel may be launched in several shapes in one bench run (the short export leg, the 1 h device leg): only the
    # dispatches with the kernel's LARGEST grid are aggregated (216 000-frame STFT launches, full 64-frame fill batches)
    rows = []
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is not None:
                    rows.append((k, int(float(row.get("Grid_Size", 0) or 0)), row["Counter_Name"],
                                 float(row["Counter_Value"])))
    biggest = defaultdict(int)
    for k, grid, _, _ in rows:
        biggest[k] = max(biggest[k], grid)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for k, grid, name, value in rows:
        if grid == biggest[k]:
            a = agg[k][name]
            a[0] += value
            a[1] += 1
    out = {}
    for k, counters in sorted(agg.items()):
        out[k] = {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in sorted(counters.items())}
    print(json.dumps(out, indent=1))
    # per-launch HBM traffic for bench.py's roofline objects: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB;
    # on gfx950 FETCH_SIZE tallies 64 B per 128-B request on wide streaming reads, so it is doubled
    # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane stores.
    def traffic(k):
        c = out.get(k, {})
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None
        return int(round(2 * 1024 * c["FETCH_SIZE"]["mean_per_dispatch"] + 1024 * c["WRITE_SIZE"]["mean_per_dispatch"]))
    tj = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py "
                    "--seconds 4 --steps 0 --cpu-frames 0 --pipe-frames 0 --device-steps 1 --device-warmup 0 --batch 128` (tools/pmc_passes.sh; largest-grid dispatches of each kernel); bytes = 2*1024*FETCH_SIZE + 1024*WRITE_SIZE",
          "raster_fill": {"config": os.environ.get("AUR_PMC_FILL_CONFIG", "3840x2160"),
                          "frames_per_launch": int(os.environ.get("AUR_PMC_FILL_FRAMES", "64")),
                          "kernel": "raster_fill_lin_kernel",
                          "hbm_bytes_per_launch": traffic("raster_fill_lin_kernel")},
          "stft": {"frames": int(os.environ.get("AUR_PMC_STFT_FRAMES", "216000")),
                   "hbm_bytes_per_launch": traffic("stft_frames_kernel")}}
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
        json.dump(tj, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
