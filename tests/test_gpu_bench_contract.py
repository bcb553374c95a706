"""bench.py's output contract, on a small workload: ONE JSON line on stdout with the fields the driver reads (metric, value,
unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the
`roofline`, `cpu_baseline` and `pipe_leg` objects; and `python bench.py --gpus 2` WITHOUT a launcher starts its own ranks
(two gloo ranks sharing the one card here) and still prints one line, with both ranks' frames counted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--seconds", "4", "--device-seconds", "8", "--steps", "1", "--warmup", "0", "--width", "640", "--height", "368",
         "--cpu-frames", "24", "--cpu-procs", "2", "--pipe-frames", "96", "--device-steps", "1", "--device-warmup", "0"]


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], cwd=REPO, env=e, capture_output=True,
                         text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_fields():
    d = _run(SMALL)
    assert d["metric"].startswith("exported frames/sec") and d["unit"] == "frames/s" and d["n_gpus"] == 1
    assert d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 240 / (d["ms_per_step"] / 1e3)) < 0.02 * d["value"]      # 4 s at 60 fps
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert r["algorithmic_bytes_per_frame"] == 640 * 368 * 3 and "traffic" in r and r["kernel"].startswith("raster_fill_lin_kernel")
    s = d["roofline_stft"]
    assert s["algorithmic_bytes_per_frame"] == 10508 and s["frames_per_launch"] == 480 and s["kernel"].startswith("stft_frames_kernel")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "frames/s" and c["value"] > 0 and "sample" in c
    assert c["piped"]["value"] > 0 and c["all_cores"]["cores"] == 2
    p = d["pipe_leg"]
    assert p["frames"] == 96 and p["rgb24"]["fps"] > 0 and p["rgb24_plain_write"]["spliced_frac"] == 0.0
    assert p["rgb24"]["bytes_per_frame"] == 640 * 368 * 3 and p["yuv420p"]["bytes_per_frame"] == 640 * 368 * 3 // 2
    assert d["roofline_pcie"]["bound"] == "pcie" and d["device_fps"] > 0 and d["stft_gbs"] > 0


def test_two_ranks_without_a_launcher():
    d = _run(["--gpus", "2", *SMALL], env={"BENCH_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "time/track shards x2" and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 240 / (d["ms_per_step"] / 1e3)) < 0.02 * d["value"]                    # both ranks' frames
