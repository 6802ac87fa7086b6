#!/usr/bin/env python3
"""Condensed view of bench.py's JSON line: throughput, per-stage times under overlap and isolated."""
import json
import sys

for path in sys.argv[1:]:
    line = json.loads(open(path).read().strip().splitlines()[-1])
    iso = line.get("stages_isolated", {})
    print(f"{path}: {line['value']:.0f} {line['unit']}  {line['ms_per_step']} ms/step  n_gpus={line['n_gpus']}")
    for s, v in line["stages"].items():
        i = iso.get(s, {})
        print(f"   {s:7s} {v['ms_per_step']:.4f} ms   isolated {i.get('ms_per_step')} ms  {i.get('algorithmic_GBps')} GB/s  hbm_frac {i.get('hbm_frac')}")
    r = line["roofline"]
    print(f"   roofline {r['kernel']}: {r['achieved']} {r['unit']} frac {r['frac']} avg_ms {r['avg_ms']} traffic {r['traffic']}")
