#!/usr/bin/env python3
"""Summarise the CSV of tools/native/prefill_phase_probe (one line per workgroup: entry, first
tile landed, unmasked tiles done, end; tile count; hardware id): time before the first tile,
time per tile, and how many of a CU's two workgroup slots are occupied over the launch.

    ./prefill_phase_probe 16 1024 32 8 > probe.csv;  python tools/prefill_phase_summary.py probe.csv
"""
import csv
import statistics
import sys
from collections import defaultdict

TICK_US = 0.01                       # s_memrealtime: 100 MHz


def main():
    for path in sys.argv[1:]:
        rows = [r for r in csv.DictReader(open(path))]
        ent = [int(r["t_entry"]) for r in rows]
        t0 = min(ent)
        end = max(int(r["t_end"]) for r in rows)
        print(f"{path}: {len(rows)} workgroups, span {(end - t0) * TICK_US:.1f} us")
        pro = [(int(r["t_first_tile"]) - int(r["t_entry"])) * TICK_US for r in rows]
        pro.sort()
        print(f"  before the first tile: mean {statistics.mean(pro):.2f} us, p50 {pro[len(pro) // 2]:.2f}, "
              f"p90 {pro[int(len(pro) * 0.9)]:.2f}")
        by_n = defaultdict(list)
        for r in rows:
            n = int(r["n_tiles"])
            by_n[n].append(((int(r["t_unmasked_done"]) - int(r["t_first_tile"])) * TICK_US,
                            (int(r["t_end"]) - int(r["t_unmasked_done"])) * TICK_US,
                            (int(r["t_end"]) - int(r["t_entry"])) * TICK_US))
        for n in sorted(by_n):
            v = by_n[n]
            um = statistics.mean(x[0] for x in v)
            dg = statistics.mean(x[1] for x in v)
            tot = statistics.mean(x[2] for x in v)
            per = um / max(n - 2, 1)
            print(f"  last pass of {n:3d} tiles: {len(v):5d} workgroups, unmasked {um:7.2f} us ({per:.2f} per tile), "
                  f"diagonal tiles + epilogue {dg:5.2f} us, workgroup total {tot:7.2f} us")
        # slot occupancy: workgroup-time per CU over the span (hw_id: cu_id bits 8-11, sh 12, se 13-15; xcc separately)
        per_cu = defaultdict(float)
        for r in rows:
            hw = int(r["hw_id"])
            cu = (int(r["xcc_id"]) & 0xF, (hw >> 8) & 0xFF)
            per_cu[cu] += (int(r["t_end"]) - int(r["t_entry"])) * TICK_US
        span = (end - t0) * TICK_US
        occ = [v / span for v in per_cu.values()]
        print(f"  {len(per_cu)} CUs seen; workgroup slots occupied over the span: mean {statistics.mean(occ):.2f} of 2 "
              f"(min {min(occ):.2f}, max {max(occ):.2f})")


if __name__ == "__main__":
    main()
