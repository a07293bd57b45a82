#!/usr/bin/env python
"""profiles/traffic.json from the PMC passes of tools/profile_configs.sh.

    python tools/pmc_traffic.py gpurun_out/<tag> <code state> [--copy-to profiles/<prefix>]

Per BASELINE config: HBM bytes of ONE step = sum over the histogram kernels of the step (the xhist:: kernels that are
not output zeroing / table building) of  2 x FETCH_SIZE + WRITE_SIZE  (KB, averaged over the dispatches of the
pass).  The factor 2 is the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies the 128-byte
requests of a wide streaming read at 64 bytes; WRITE_SIZE is taken as reported.  bench.py reports the figure as
`roofline.traffic` only when the kernel description and the samples per launch of its own run match the entry."""
import json
import os
import re
import shutil
import sys


def counters(path):
    """{kernel name: (dispatches, avg)} from the '## PMC counters' section of a rocpd_summary text"""
    out, on = {}, False
    for line in open(path):
        if line.startswith("## PMC counters"):
            on = True
            continue
        if on and "|" in line:
            f = [x.strip() for x in line.split("|")]
            out[f[0]] = (int(f[2]), float(f[3]))
    return out


def steady_averages(path):
    """{kernel name: (launches, skipped, avg_us)} from the '## xhist kernels after their first' section of a rocpd_summary text"""
    out, on = {}, False
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("## xhist kernels after their first"):
            on = True
            continue
        if on and line.startswith("##"):
            break
        if on and "|" in line:
            f = [x.strip() for x in line.split("|")]
            out[f[0]] = (int(f[1]), int(f[2]), float(f[3]))
    return out


def hist_kernels(d):
    skip = ("zero_words", "build_tables", "minmax", "buffer_add")
    return {k: v for k, v in d.items() if "xhist::" in k and not any(s in k for s in skip)}


def main():
    src, state = sys.argv[1], sys.argv[2]
    copy_to = sys.argv[sys.argv.index("--copy-to") + 1] if "--copy-to" in sys.argv else None
    tpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    doc = {"configs": {}}
    if os.path.exists(tpath):
        try:
            old = json.load(open(tpath))
            if "configs" in old:
                doc = old
        except Exception:
            pass
    doc["correction"] = "gfx950: FETCH_SIZE counts a 16 B/lane streaming read at half its bytes (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE as reported"
    for fn in sorted(os.listdir(src)):
        m = re.match(r"(c\du?(?:_full)?)_pmc_FETCH_SIZE\.txt$", fn)
        if not m:
            continue
        name = m.group(1)
        fetch = hist_kernels(counters(os.path.join(src, fn)))
        write = hist_kernels(counters(os.path.join(src, "%s_pmc_WRITE_SIZE.txt" % name)))
        bench = json.load(open(os.path.join(src, "%s_bench_under_pmc_FETCH_SIZE.json" % name)))
        total, per_kernel = 0.0, {}
        for k, (nd, f_kb) in fetch.items():
            w_kb = write.get(k, (0, 0.0))[1]
            # launches of this kernel per step (a line with "first_call_ms" timed one more launch ahead of the warm-up)
            per_step = nd / float(bench["steps"] + bench["warmup"] + (1 if "first_call_ms" in bench else 0))
            b = (2.0 * f_kb + w_kb) * 1024.0 * per_step
            per_kernel[re.sub(r"\(.*", "", k)[:80]] = {"fetch_KB_avg": f_kb, "write_KB_avg": w_kb, "launches_per_step": per_step, "hbm_bytes_per_step": b}
            total += b
        alg = bench["roofline"]["algorithmic_bytes_per_launch"]
        # rocprofv3 --kernel-trace averages of the same command, first launches left out: summed over the histogram kernels
        # of a step (those the counters saw); and the HIP-event mean bench.py printed INSIDE that profiled process
        steady = hist_kernels(steady_averages(os.path.join(src, "%s_kernel_stats.txt" % name)))
        steady = {k: v for k, v in steady.items() if "build_pack_tables" not in k and "gather_rows" not in k}
        rocprof_avg_us = sum(v[2] for v in steady.values()) if steady else None
        skipped = max((v[1] for v in steady.values()), default=None)
        try:
            under = json.load(open(os.path.join(src, "%s_bench_under_rocprof.json" % name)))["roofline"]["kernel_ms_mean"]
        except Exception:
            under = None
        doc["configs"][name] = {
            "rocprof_avg_us": rocprof_avg_us,
            "rocprof_launches_skipped": skipped,
            "rocprof_kernels": {re.sub(r"\(.*", "", k)[:80]: {"launches": v[0], "avg_us": v[2]} for k, v in steady.items()},
            "events_ms_under_rocprof": under,
            "kernel": bench["config"]["kernel"],
            "samples_per_launch": bench["config"]["samples_per_gpu"],
            "hbm_bytes_per_launch": total,
            "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": total / alg,
            "per_kernel": per_kernel,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --config %s%s`, summaries %s_pmc_*.txt"
                      % (name[:2] + (" --unweighted" if name[2:3] == "u" else ""), " --full" if name.endswith("_full") else "", (os.path.basename(copy_to) + "_" if copy_to else "") + name),
            "code_state": state,
            # hash of xhistogram_amd/csrc as the measured process saw it (bench.py::csrc_sha16): bench.py reports these counters
            # only from a library built from the same sources
            "csrc_sha16": bench.get("roofline", {}).get("csrc_sha16"),
        }
        print("%-8s traffic %.4g B / algorithmic %.4g B = %.4f" % (name, total, alg, total / alg))
    json.dump(doc, open(tpath, "w"), indent=1)
    if copy_to:
        for fn in sorted(os.listdir(src)):
            if fn.endswith((".txt", ".json")) and not fn.endswith(".err"):
                shutil.copy(os.path.join(src, fn), "%s_%s" % (copy_to, fn))


if __name__ == "__main__":
    main()
