#!/usr/bin/env python3
"""VALU instruction count of the STFT frame loop -> profiles/stft_isa.json (read by bench.py for the issue roof).

    python tools/stft_isa.py        # CPU only: hipcc -S of csrc/stft.hip + tools/isa_hist.py on the headline instantiation
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("STFT_ISA_KERNEL", "stft_frames_kernelILi2ELb1ELb0ELb0ELb1E")   # stereo, log-spectrum stored, no band maxima, no sample output, zero tail (hop <= 896: the 48 kHz / 60 fps headline)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "stft.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                               "-fPIC", "-fno-slp-vectorize", "--cuda-device-only", "-S", os.path.join(ROOT, "aurora_amd", "csrc", "stft.hip"),
                               "-o", asm], stderr=subprocess.DEVNULL)
        txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "isa_hist.py"), asm, KERNEL, "60"],
                                      text=True)
        vg = re.search(r"\.vgpr_count:\s+(\d+)", open(asm).read().split(KERNEL)[-1])
    m = re.search(r"total (\d+)\s+valu (\d+)\s+f64 (\d+)\s+lds (\d+)", txt)
    # the frame loop is unrolled by hand (ping-pong of the two block arrays): one frame = 64 ds_write_b64 of the exchanges
    w = re.search(r"^\s*(\d+) ds_write_b64", txt, re.M)
    frames = max(1, int(w.group(1)) // 64) if w else 1
    out = {"kernel": KERNEL, "frames_per_loop_iteration": frames, "loop_instructions": int(m.group(1)),
           "valu_per_frame": int(m.group(2)) // frames, "f64_per_frame": int(m.group(3)) // frames,
           "lds_per_frame": int(m.group(4)) // frames,
           "vgprs": int(vg.group(1)) if vg else None, "histogram": txt.strip().split("\n")[1:]}
    path = os.path.join(ROOT, "profiles", "stft_isa.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(txt.split("\n")[0], "->", path)


if __name__ == "__main__":
    main()
