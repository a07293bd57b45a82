#!/usr/bin/env python
"""Reduce a discovery run of tests/test_gpu_census.py to the cases worth keeping (greedy set cover).

    XHIST_CENSUS_DISCOVER=gpurun_out/census/discover.jsonl XHIST_AMD_KERNEL_LOG_ALL=1 XHIST_AMD_KERNEL_LOG=... \
        python -m pytest tests/test_gpu_census.py -m gpu -q
    python tools/census_cover.py gpurun_out/census/discover.jsonl [--already other_tests.log] > tests/golden/census_cases.json

Every line of the discovery file is {"case": key, "kernels": [symbols the library picked during that case]}.  Kernels that the
rest of the GPU suite selects anyway (--already: its kernel log) need no case here."""
import json
import sys


def main():
    argv = sys.argv[1:]
    already = set()
    if "--already" in argv:
        i = argv.index("--already")
        already = {l.strip() for l in open(argv[i + 1]) if l.strip()}
        del argv[i:i + 2]
    picks = {}
    for line in open(argv[0]):
        d = json.loads(line)
        picks.setdefault(d["case"], set()).update(k for k in d["kernels"] if k and k != "?")
    todo = set().union(*picks.values()) - already
    chosen = []
    while todo:
        best = max(picks, key=lambda c: (len(picks[c] & todo), -len(c)))
        gain = picks[best] & todo
        if not gain:
            break
        chosen.append(best)
        todo -= gain
    json.dump({"what": "cases of tests/test_gpu_census.py that together select every kernel its whole product selects (tools/census_cover.py)",
               "kernels_covered": len(set().union(*picks.values()) - already), "cases_in_product": len(picks), "cases": sorted(chosen)}, sys.stdout, indent=0)
    print()


if __name__ == "__main__":
    main()
