#!/bin/bash
# Rates of every BASELINE.json configuration (and the closed-path modes) on one MI355X: frames/s reaching the host sink
# (`value`) and the device-side rate with the frames left in HBM (`device_fps`).  The table in DESIGN.md section 6.
# configs[0] is a 30 s track (900 frames, two launches): 40 steps of it; a short device leg right after the PCIe-bound
# export leg (GPU 98 % idle) reads low while the clocks come up.
# gpurun: bash tools/config_table.sh > gpurun_out/config_table.txt
cd "$(dirname "$0")/.."
show() { grep -h '^{"metric"' "$1" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; p=d["roofline_pcie"]; q=(d.get("pipe_leg") or {}).get("rgb24") or {}; print("  host sink %.0f frames/s (%.1f GB/s)  one pipe %s frames/s (%s GB/s)  device %.0f frames/s  %s %.0f GB/s = %.3f of HBM peak, %.4f ms/launch of %d frames" % (d["value"], p["achieved"], q.get("fps"), q.get("gbs"), d["device_fps"], r["kernel"], r["achieved"], r["frac"], r["avg_launch_ms"], r["frames_per_launch"]))'; }
run() { echo "== $1"; shift; timeout -k 10 200 python3 -u bench.py "$@" --steps 1 --warmup 1 --cpu-frames 0 --yuv-leg 0 --device-steps 2 > gpurun_out/cfg.tmp 2>&1 || { tail -5 gpurun_out/cfg.tmp; exit 1; }; show gpurun_out/cfg.tmp; }
run "configs[0] 30 s 44.1 kHz Spectrum - Linear 1280x720@30" --signal sweep --seconds 30 --sample-rate 44100 --fps 30 --width 1280 --height 720 --device-seconds 30 --device-warmup 10 --device-steps 40
run "configs[1] 240 s Spectrum - Linear 1920x1080@60" --seconds 240 --width 1920 --height 1080 --device-seconds 240
run "configs[2] 240 s Waveform - Linear 3840x2160@60" --seconds 240 --mode "Waveform - Linear" --device-seconds 240
run "configs[3] 240 s Particles + Feather 1920x1080@60" --seconds 240 --mode "Particles + Feather" --width 1920 --height 1080 --device-seconds 240
run "configs[4] unit: 3600 s Spectrum - Linear 3840x2160@60 (host-sink leg over the first 240 s)" --seconds 240
run "Spectrum - Radial 3840x2160@60 240 s" --seconds 240 --mode "Spectrum - Radial" --device-seconds 240
run "Waveform - Circular 3840x2160@60 240 s" --seconds 240 --mode "Waveform - Circular" --device-seconds 240
run "Spectrum - Radial + glow 3840x2160@60 240 s" --seconds 240 --mode "Spectrum - Radial" --glow --device-seconds 240
