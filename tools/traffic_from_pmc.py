#!/usr/bin/env python3
"""profiles/traffic.json from the committed rocprofv3 PMC summaries (profiles/r<N>_<cfg>_pmc.txt, written by
tools/collect_profiles.sh + tools/pmc_summary.py): HBM-side bytes per launch of each configuration's dominant
kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  The factor 2 is the gfx950 correction of
MI355X_MICROARCH.md ("FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read",
16 B per lane -- the (c,w) stream of the chain kernels and the scan of the online kernels are such reads;
the counter also tallies Infinity-Cache hits, so this is memory-side traffic, an upper bound of HBM bytes).
Also, from the SQ pass of the same file: `valu_issue_busy` = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs over the launch's
cycles (GRBM_GUI_ACTIVE / 8 XCDs) and `sustained_clock_ghz` = those cycles over the launch's duration in that pass.
bench.py quotes these figures as `roofline.traffic` / `valu_issue_busy` / `sustained_clock_ghz` with their source."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r4"
# key = "<bench.py --config>_<--arith>" (what bench.py looks up); value = (summary tag, dominant kernel)
KERNEL = {"c3_strict": ("c3", r"vsom_update_std_nt4_gfx950"), "c3_sigma": ("c3_sigma", r"vsom_update_sfma_nt4_gfx950"),
          "c3_contracted": ("c3_contracted", r"vsom_update_fma_nt4_gfx950"),
          "c2_strict": ("c2", r"vsom_update_std_nt4_gfx950"), "c4_strict": ("c4", r"update_chain3_kernel"),
          "c5_strict": ("c5", r"vsom_update_clr_rp8_gfx950"), "online_strict": ("online", r"onl_fused_kernel")}


def counters(path, kernel):
    """{counter: value} of the first pass that carries it, plus "<counter>@us" = the kernel's duration in that pass"""
    vals, cur, us = {}, None, 0.0
    for ln in open(path):
        if not ln.startswith(" ") and not ln.startswith("#"):
            cur = ln.split("  dispatches=")[0].strip()
            m = re.search(r"avg_us=([0-9.]+)", ln)
            us = float(m.group(1)) if m else 0.0
        elif cur and re.search(kernel, cur):
            parts = ln.split()
            if len(parts) == 2 and parts[0] not in vals:
                vals[parts[0]] = float(parts[1])
                vals[parts[0] + "@us"] = us
    return vals


out = {"_doc": __doc__.strip().replace("\n", " ")}
for cfg, (tag, kern) in KERNEL.items():
    p = os.path.join(PROF, f"{ROUND}_{tag}_pmc.txt")
    if not os.path.exists(p):
        continue
    v = counters(p, kern)
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    ent = {"kernel": kern, "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"],
           "hbm_bytes_per_launch": int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0)),
           "source": f"profiles/{ROUND}_{tag}_pmc.txt, separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE"}
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        ent["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
        ent["valu_issue_busy"] = round(v["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc, 4)
        # (launches of a few microseconds: the counter window and the traced duration do not cover the same interval)
        if v.get("GRBM_GUI_ACTIVE@us", 0) > 100:
            ent["sustained_clock_ghz"] = round(cyc / v["GRBM_GUI_ACTIVE@us"] / 1e3, 3)
            ent["clock_note"] = "cycles of the launch (GRBM_GUI_ACTIVE / 8) over its duration under the counter pass"
    if "TCC_HIT" in v and "TCC_MISS" in v:
        ent["l2_hit_rate"] = round(v["TCC_HIT"] / (v["TCC_HIT"] + v["TCC_MISS"]), 4)
    out[cfg] = ent
json.dump(out, open(os.path.join(PROF, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
