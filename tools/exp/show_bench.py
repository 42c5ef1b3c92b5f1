#!/usr/bin/env python3
"""print the figures of bench.py JSON lines that matter when comparing runs"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, "value", d["value"], "ms", d["ms_per_step"], d["kernel_ms_per_step"], "frac", d["roofline"].get("frac"), d["roofline"].get("frac_executed"), d.get("bmu_shortlist_last"))
    for o in d.get("other_arithmetics", []):
        print("    ", o["arithmetic"], o["ms_per_step"])
    for v in d.get("data_variants", []):
        print("    ", v["data"], v["ms_per_step"], "vs", v["vs_headline"], "bmu", v["bmu_ms"], "upd", v["update_ms"], v["roofline"]["frac"], v["roofline"]["frac_executed"], v["bmu_shortlist_last"])
