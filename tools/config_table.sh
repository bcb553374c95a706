#!/bin/bash
# Device-side rate of every BASELINE.json configuration on one MI355X (the table in DESIGN.md section 6).
cd "$(dirname "$0")/.."
show() { grep -h '^{"metric"' "$1" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("  frames/s %.0f  ms/step %.2f  %s %.1f GB/s  %.4f ms/launch" % (d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["avg_launch_ms"]))'; }
run() { echo "== $1"; shift; timeout -k 10 150 python -u bench.py "$@" --steps 3 --cpu-frames 0 --pcie-frames 0 > gpurun_out/cfg.tmp 2>&1 || { tail -5 gpurun_out/cfg.tmp; exit 1; }; show gpurun_out/cfg.tmp; }
run "configs[0] 30 s 44.1 kHz Spectrum - Linear 1280x720@30" --seconds 30 --sample-rate 44100 --fps 30 --width 1280 --height 720
run "configs[1] 240 s Spectrum - Linear 1920x1080@60" --seconds 240 --width 1920 --height 1080
run "configs[2] 240 s Waveform - Linear 3840x2160@60" --seconds 240 --mode "Waveform - Linear"
run "configs[3] 240 s Particles + Feather 1920x1080@60" --seconds 240 --mode "Particles + Feather" --width 1920 --height 1080
run "Spectrum - Radial 3840x2160@60 240 s" --seconds 240 --mode "Spectrum - Radial"
run "Waveform - Circular 3840x2160@60 240 s" --seconds 240 --mode "Waveform - Circular"
run "Spectrum - Radial + glow 3840x2160@60 240 s" --seconds 240 --mode "Spectrum - Radial" --glow
