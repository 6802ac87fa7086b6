#!/usr/bin/env python3
"""Steady-state timeline of the default schedule from a rocprofv3 kernel trace:
     rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --cpu-sample 0 --no-ingest --no-isolated
     python tools/timeline.py gpurun_out/tl/t_kernel_trace.csv
   Prints one period (from one response-kernel launch of slice 0 to the next) with start offset, duration and queue of every kernel,
   and the busy time of every queue over the period."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:24], r["Queue_Id"])
                for r in rows)
    eig = [e for e in ev if e[2].startswith("k_mineig")]
    q0 = eig[0][3]
    starts = [e[0] for e in eig if e[3] == q0]
    k = int(len(starts) * 0.7)
    t0, t1 = starts[k], starts[k + 1]
    print(f"period {(t1 - t0) / 1e3:.1f} us (launch {k} of {len(starts)} on queue {q0})")
    busy = collections.defaultdict(float)
    for s, e, n, q in ev:
        if t0 <= s < t1:
            if not n.startswith("__amd"):
                print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}")
            busy[q] += (e - s) / 1e3
    print({f"q{q}": round(v, 1) for q, v in sorted(busy.items())})


if __name__ == "__main__":
    main()
