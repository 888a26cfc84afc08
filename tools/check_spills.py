#!/usr/bin/env python3
"""ISA audit (no GPU needed): for every kernel of a HIP source, count scratch_* (spill) instructions and how
many of them sit between the first and the last v_mfma, i.e. inside the main loop.
  python tools/check_spills.py stablediffusion_amd/csrc/igemm2.hip [attention.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=fast", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-x", "hip", "--cuda-device-only", "-S"]
bad = 0
for src in sys.argv[1:]:
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, os.path.abspath(src), "-o", f.name], check=True, stderr=subprocess.DEVNULL,
                       cwd=os.path.dirname(os.path.abspath(src)) or ".")
        lines = open(f.name).read().splitlines()
    starts = [(i, l[:-1]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, (i, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        body = lines[i:end]
        mf = [j for j, l in enumerate(body) if "v_mfma" in l]
        sc = [j for j, l in enumerate(body) if "scratch_" in l]
        inside = sum(1 for j in sc if mf and mf[0] < j < mf[-1])
        vg = next((l.split()[-1] for l in body if ".amdhsa_next_free_vgpr" in l), "?")
        if sc or inside:
            short = re.sub(r"^_ZN2sd12_GLOBAL__N_1\d+", "", name)[:70]
            print(f"{os.path.basename(src)}: {short:70s} vgpr {vg:>4s} scratch ops {len(sc):3d} inside the MFMA range {inside}")
        bad += inside
print("kernels with spill traffic inside the MFMA range:", "none" if bad == 0 else bad)
sys.exit(1 if bad else 0)
