#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* / GRBM_* passes (tools/profile_sq.sh) -> per-kernel JSON for profiles/: for the heaviest kernels of
the UNet forward, MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)), wait / issue-stall shares of SQ_WAVE_CYCLES, LDS bank-conflict share, and the clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration is not available in a counter-only pass: reported as GUI cycles per launch).
usage: summarize_sq.py <dir with a/ b/> <out.json>"""
import collections
import glob
import json
import re
import sqlite3
import sys

root, out = sys.argv[1], sys.argv[2]


def load(d):
    a = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(d + "/*/*_results.db") + glob.glob(d + "/*_results.db"):
        db = sqlite3.connect(path)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
        tab = "counters_collection" if "counters_collection" in tabs else next(t for t in tabs if "counter" in t.lower())
        for name, cname, value in db.execute(f"select kernel_name, counter_name, value from {tab}"):
            a[name][cname][0] += float(value)
            a[name][cname][1] += 1
    return a


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"^sd::", "", k)
    m = re.match(r"_ZN2sd12_GLOBAL__N_1\d+([a-z_0-9]+)I(Li(\d+))?", k)
    if m:
        return m.group(1) + (f"<{m.group(3)},...>" if m.group(3) else "")
    return k.split("(")[0].replace(" ", "")


A, B = load(root + "/a"), load(root + "/b")
rows = {}
for k, c in A.items():
    if "sd" not in k and "GLOBAL" not in k:
        continue
    n = c["SQ_WAVE_CYCLES"][1] or 1
    g = lambda name, src=c: src[name][0] / max(src[name][1], 1)
    wave = g("SQ_WAVE_CYCLES")
    busy = g("SQ_BUSY_CYCLES")
    b = B.get(k, {})
    gb = lambda name: (b[name][0] / max(b[name][1], 1)) if name in b else None
    row = {"launches": n,
           "sq_busy_cycles_per_launch": round(busy),
           "grbm_gui_active_per_launch": round(g("GRBM_GUI_ACTIVE")),
           "mfma_busy_cycles_per_launch": round(g("SQ_VALU_MFMA_BUSY_CYCLES")),
           "mfma_busy_frac": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("GRBM_GUI_ACTIVE") else 0.0, 4),
           "wait_inst_any_frac_of_wave_cycles": round(g("SQ_WAIT_INST_ANY") / wave if wave else 0.0, 4),
           "wait_any_frac_of_wave_cycles": round(g("SQ_WAIT_ANY") / wave if wave else 0.0, 4),
           "wait_inst_lds_frac_of_wave_cycles": round(g("SQ_WAIT_INST_LDS") / wave if wave else 0.0, 4),
           "active_inst_any_frac_of_wave_cycles": round(g("SQ_ACTIVE_INST_ANY") / wave if wave else 0.0, 4)}
    if b:
        idx = gb("SQ_LDS_IDX_ACTIVE")
        row.update({"lds_bank_conflict_frac_of_lds_active": round(gb("SQ_LDS_BANK_CONFLICT") / idx, 4) if idx else None,
                    "mfma_coexec_cycles_per_launch": round(gb("SQ_VALU_MFMA_COEXEC_CYCLES") or 0),
                    "mfma_mops_f16_per_launch": round(gb("SQ_INSTS_VALU_MFMA_MOPS_F16") or 0)})
    row["_weight"] = busy * n
    rows[short(k)] = row
top = dict(sorted(rows.items(), key=lambda kv: -kv[1]["_weight"])[:16])
for r in top.values():
    r.pop("_weight")
res = {"_note": "rocprofv3 --pmc, two separate passes (tools/profile_sq.sh) over `python3 tools/run_unet.py --iters 2` (SD1.5 UNet, "
                "CFG batch 8, 64x64 latents); averages per launch over every launch of the kernel; the 16 heaviest kernels by "
                "SQ_BUSY_CYCLES x launches.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES) as the "
                "guide's units give it (MFMA_BUSY in cycles, summed over SIMDs; BUSY per CU) -- compare between kernels, and "
                "with the attention figure of profiles/r02_attention_ablation.txt (39 %), rather than as an absolute."}
res.update(top)
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(top), "kernels")
