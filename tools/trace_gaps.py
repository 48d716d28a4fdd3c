"""Durations of, and gaps between, the kernels of a per-scan loop, from a `rocprofv3 --kernel-trace --output-format csv` run:
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- <program> <args>      # the program itself after `--`
    python3 tools/trace_gaps.py /tmp/tr
Prints, for the second half of the trace (steady state), the mean duration per kernel and mean / p10 / p50 / p90 of the idle time
between consecutive kernels by (predecessor, successor) -- what DESIGN.md's "per-scan flow" section was worked out from."""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = [d] if os.path.isfile(d) else glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % d)
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(r["Kernel_Name"].split("(")[0][-36:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    ev = ev[len(ev) // 2:]
    durs, gaps = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in zip(ev, ev[1:]):
        durs[a[0]].append((a[2] - a[1]) / 1e3)
        gaps[(a[0], b[0])].append((b[1] - a[2]) / 1e3)
    for k, v in durs.items():
        print("kernel %-38s n=%4d mean %6.2f us" % (k, len(v), sum(v) / len(v)))
    for k, v in gaps.items():
        v = sorted(v)
        print("gap    %-38s -> %-38s n=%4d mean %6.2f  p10 %5.2f  p50 %5.2f  p90 %5.2f us" %
              (k[0], k[1], len(v), sum(v) / len(v), v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10]))


if __name__ == "__main__":
    main()
