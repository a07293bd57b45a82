#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) result: per-kernel stats and PMC counters.

    python tools/rocpd_summary.py gpurun_out/prof_stats/bench_results.db > profiles/<name>.txt
"""
import re
import sqlite3
import sys


SKIP = 6  # launches left out of the steady-state averages


def short(name, n=110):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= n else name[: n - 3] + "..."


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print("# rocprofv3 summary of", path)
    print("## kernel stats (name | calls | total_us | avg_us | min_us | max_us | %)")
    rows = list(cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    for r in rows:
        print("%-112s | %5d | %12.1f | %10.2f | %10.2f | %10.2f | %5.2f" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    # the dominant xhist kernel launch by launch, in dispatch order: where inside a run the slow launches sit (VERDICT r3 "weak" #2:
    # "one in 23 launches at +27 %")
    top = [r for r in rows if "xhist::" in r[0] and "zero_words" not in r[0] and "build" not in r[0]]
    if top:
        try:
            series = [d for (d,) in cur.execute("select duration/1e3 from kernels where name = ? order by start", (top[0][0],))]
            print("## launches of the dominant kernel in dispatch order, us (%d): %s" % (len(series), " ".join("%.0f" % d for d in series[:64])))
        except sqlite3.Error:
            pass
    # every xhist histogram kernel without its first launches: after the idle gap of data generation the first ~6 launches
    # of a burst run up to 14 % slow (clock excursion), and a --stats average over warm-up + timed launches
    # carries them; bench.py's HIP-event mean does not (10 untimed launches first)
    if top:
        print("## xhist kernels after their first %d launches (name | launches | skipped | avg_us | min_us | max_us)" % SKIP)
        for r in top:
            try:
                series = [d for (d,) in cur.execute("select duration/1e3 from kernels where name = ? order by start", (r[0],))]
            except sqlite3.Error:
                continue
            skip = SKIP if len(series) > 2 * SKIP else 0
            rest = series[skip:]
            print("%-112s | %5d | %3d | %10.2f | %10.2f | %10.2f" % (short(r[0]), len(series), skip, sum(rest) / len(rest), min(rest), max(rest)))
    k = list(cur.execute(
        "select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels where name like '%xhist::%' group by name, grid_x, workgroup_x"))
    if k:
        print("## xhist dispatch geometry (name | grid | workgroup | lds | vgpr | agpr | sgpr | scratch)")
        for r in k:
            print("%-112s | %s" % (short(r[0]), " | ".join(str(v) for v in r[1:])))
    try:
        c = list(cur.execute(
            "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection group by kernel_name, counter_name order by 1, 2"))
    except sqlite3.Error:
        c = []
    if c:
        print("## PMC counters (kernel | counter | dispatches | avg | min | max)")
        for r in c:
            print("%-112s | %-18s | %4d | %16.3f | %16.3f | %16.3f" % (short(r[0]), r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1])
