#!/usr/bin/env python3
"""print the headline figures of a `bench.py --config online` line read from stdin (label = argv[1])"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "samples/s", d["roofline"].get("avg_sample_us"), "us/sample",
      d.get("online_search"), flush=True)
