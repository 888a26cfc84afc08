#!/usr/bin/env python3
"""Register / scratch usage per kernel from a hipcc -S listing: tools/kernel_regs.py file.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = re.compile(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n'
                 r'(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)')
for m in pat.finditer(s):
    n = m.group(1)
    try:
        dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        dn = n
    dn = dn.replace("sd::(anonymous namespace)::", "").split("(")[0]
    if flt in dn:
        print(f"{dn[:80]:80s} scratch {m.group(2):>4s} sgpr {m.group(3):>3s} vgpr {m.group(4):>3s} spill {m.group(5)}")
