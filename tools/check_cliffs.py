#!/usr/bin/env python
"""Regression gate over the shape scanners (VERDICT r3 "next" #7).

xhist_exec_device.hip.h picks kernels and launch geometry from ~15 hand-fitted thresholds; a change that moves one of them can
push a class of shapes off a cliff without touching any BASELINE config.  The three scanners cover the space those thresholds
cut up:   tools/shape_cliffs.py (rows x cols x bins x dtype),  tools/size_ramp.py (10^5 ... 10^9 samples of one row),
tools/hist2d_sizes.py (64^2 ... 1024^2 bins x 10^5 ... 10^8 samples).

  python tools/check_cliffs.py run <dir>                       run the scanners on this GPU, write <dir>/*.jsonl
  python tools/check_cliffs.py compare <baseline dir> <dir>    exit 1 if a cell is more than --tol (12 %) slower than the baseline

Boxes differ by a few percent in HBM rate and clocks, so `compare` first takes the MEDIAN ratio new / baseline over all cells
(the box factor) and judges every cell against it; cells under 20 us get 2 us of slack (launch jitter).  Noise floor: the same
code scanned three times on ONE box differs by up to 11 % in one cell of 122 per pair of scans (mid-size kernels, whose clocks
move with what ran before them) even though every measurement now warms up for 25 ms first — hence 12 %, not 10.  The committed
baseline is profiles/cliffs_baseline/ (its README line says which commit and when); tools/profile_configs.sh runs the gate
at the end of every evidence set.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCANNERS = ("shape_cliffs", "size_ramp", "hist2d_sizes")


def cells(path, scanner):
    """{cell name: milliseconds} of one scanner's JSON lines"""
    out = {}
    for line in open(path):
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if scanner == "shape_cliffs":
            kind = d.get("edges", "linspace")
            out["%dx%d %s bins=%s w=%d%s" % (d["rows"], d["cols"], d["dtype"], "x".join(map(str, d["bins"])), d["weighted"],
                                              "" if kind == "linspace" else " edges=" + kind)] = d["ms"]
        elif scanner == "size_ramp":
            out["%s n=%d" % (d["case"], d["n"])] = d["us"] / 1e3
        else:
            for n, v in d["us"].items():
                out["2-D %dx%d w=%d n=%s" % (d["nb"], d["nb"], d["weighted"], n)] = v[0] / 1e3
    return out


def run(outdir, repeats=2):
    """every scanner `repeats` times (<name>.jsonl, <name>.2.jsonl ...): compare() takes a cell's MINIMUM over the repeats —
    some mid-size cells are bimodal from one process to the next on the same box and code (f64 + weights, 10^8 samples:
    0.231 or 0.259 ms), and the slow mode is not a property of the code under test"""
    os.makedirs(outdir, exist_ok=True)
    for rep in range(repeats):
        for s in SCANNERS:
            name = s + (".jsonl" if rep == 0 else ".%d.jsonl" % (rep + 1))
            with open(os.path.join(outdir, name), "w") as f:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", s + ".py")], stdout=f, stderr=subprocess.PIPE, text=True, timeout=1800)
            if r.returncode:
                sys.stderr.write(r.stderr[-2000:])
                raise SystemExit("scanner %s failed (rc %d)" % (s, r.returncode))
    try:
        state = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "snapshot (no .git on this box)"
    except OSError:
        state = "snapshot"
    with open(os.path.join(outdir, "README"), "w") as f:
        f.write("cliff-scanner cells of code state %s\n" % state)


def compare(base_dir, new_dir, tol):
    base, new = {}, {}
    for s in SCANNERS:
        for d, into in ((base_dir, base), (new_dir, new)):
            for name in sorted(os.listdir(d)):
                if name == s + ".jsonl" or (name.startswith(s + ".") and name.endswith(".jsonl")):
                    for k, v in cells(os.path.join(d, name), s).items():
                        key = s + ": " + k
                        into[key] = min(into.get(key, v), v)  # the minimum over the repeats of a scan
    common = sorted(set(base) & set(new))
    if not common:
        raise SystemExit("no common cells between %s and %s" % (base_dir, new_dir))
    ratios = sorted(new[k] / base[k] for k in common if base[k] > 0)
    box = ratios[len(ratios) // 2]
    bad, better = [], []
    for k in common:
        want = base[k] * box
        slack = 0.002 if base[k] < 0.020 else 0.0
        if new[k] > want * (1 + tol) + slack:
            bad.append((new[k] / want, k, base[k], new[k]))
        elif new[k] < want * (1 - tol) - slack:
            better.append((new[k] / want, k, base[k], new[k]))
    print("%d cells in common (%d only in the baseline, %d new); box factor (median new / baseline) %.3f; tolerance %.0f %%"
          % (len(common), len(set(base) - set(new)), len(set(new) - set(base)), box, tol * 100))
    for title, rows in (("SLOWER", sorted(bad, reverse=True)), ("faster", sorted(better))):
        for r, k, b, n in rows:
            print("  %s  x%.2f  %-60s %.4f -> %.4f ms" % (title, r, k, b, n))
    print("gate: %s" % ("FAIL: %d cell(s) regressed" % len(bad) if bad else "ok"))
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "compare":
        tol = 0.12
        if "--tol" in sys.argv:
            tol = float(sys.argv[sys.argv.index("--tol") + 1])
        raise SystemExit(compare(sys.argv[2], sys.argv[3], tol))
    else:
        raise SystemExit(__doc__)
