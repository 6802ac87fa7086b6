#!/usr/bin/env python3
"""Compact view of a rocprofv3 *_kernel_stats.csv:  python tools/kstats.py <csv>"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:28]:28s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  min {int(r['MinNs']) / 1e3:9.1f}  max {int(r['MaxNs']) / 1e3:9.1f}  {r['Percentage']}%")
