#!/usr/bin/env python3
"""Summarise a rocprofv3 ``*_kernel_trace.csv``: per-kernel time inside a window.

    python tools/prof_summary.py TRACE.csv [--after KERNEL_SUBSTR] [--last-steps N --marker SUBSTR]

``--after X`` keeps only dispatches that start after the LAST dispatch whose name contains X
(e.g. ``flash_prefill`` to isolate the decode phase).  ``--marker M --last-steps N`` keeps the
dispatches of the last N "steps", where a step ends with every occurrence of marker kernel M
(use the sampler kernel: one per step); ``--marker M --skip-steps S --first-steps N`` keeps steps
S .. S+N-1 counted from the start of the trace (S >= 1).
Prints busy time, wall span and the idle share (launch gaps) of the window.
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(Custom_)?Cijk_.*?(MT\d+x\d+x\d+)", name)
    if m:
        return "hipBLASLt GEMM " + m.group(2)
    return name.split("(")[0][:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--after", default=None)
    ap.add_argument("--marker", default=None)
    ap.add_argument("--last-steps", type=int, default=0)
    ap.add_argument("--first-steps", type=int, default=0,
                    help="with --marker: the N steps that follow the first --skip-steps steps")
    ap.add_argument("--skip-steps", type=int, default=0)
    ap.add_argument("--top", type=int, default=14)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if a.after:
        idx = max(i for i, r in enumerate(rows) if a.after in r[2])
        rows = rows[idx + 1:]
    if a.marker and (a.last_steps or a.first_steps):
        marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
        if a.first_steps:        # step k ends at marks[k]: steps skip .. skip + n - 1
            marks = marks[max(a.skip_steps - 1, 0):a.skip_steps + a.first_steps]
        else:
            marks = marks[-(a.last_steps + 1):]
        rows = rows[marks[0] + 1: marks[-1] + 1]
        steps = len(marks) - 1
    else:
        steps = 1
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = short(n)
        agg[k][0] += e - s
        agg[k][1] += 1
    print(f"window: {len(rows)} dispatches, {steps} step(s), span {span/1e6/steps:.3f} ms/step, "
          f"busy {busy/1e6/steps:.3f} ms/step, idle {100*(1-busy/span):.1f}%")
    print(f"{'kernel':72s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'share':>6s}")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{k:72s} {c/steps:10.1f} {t/1e6/steps:9.3f} {t/1e3/c:9.1f} {100*t/busy:5.1f}%")


if __name__ == "__main__":
    main()
