#!/usr/bin/env python
"""Which of the library's kernel instantiations does anything select?   (VERDICT r4 "next" #7)

    XHIST_AMD_KERNEL_LOG=/path/log  <any workload: pytest -m gpu, tools/check_cliffs.py run, tools/soak.py ...>
    python tools/kernel_census.py xhistogram_amd/libxhist_amd.so /path/log [more logs] [--unused] [--json out.json]

The library appends the name of every distinct kernel it launches through a dispatch table to $XHIST_AMD_KERNEL_LOG
(csrc/xhist_host_common.hip.h::log_picked_kernel).  This tool lists the instantiations the shared object holds (the host
stubs `__device_stub__<kernel>` of its symbol table), groups both by kernel template, and prints per template how many
instantiations exist and how many were ever selected."""
import collections
import json
import re
import subprocess
import sys


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def norm(name):
    m = re.match(r"(_ZN5xhist)(\d+)__device_stub__(.*)$", name.strip())
    if m:  # a symbol c++filt / nm could not demangle (_Float16 arguments): fix the length prefix by hand
        name = "%s%d%s" % (m.group(1), int(m.group(2)) - len("__device_stub__"), m.group(3))
    name = name.replace("__device_stub__", "")
    name = re.sub(r"^void ", "", name.strip())
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    depth, cut = 0, len(name)  # drop the parameter list: the last top-level '('
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].strip()


def template_of(name):
    m = re.match(r"_ZN5xhist\d+([a-z_0-9]+?)I", name)
    if m:
        return "xhist::" + m.group(1) + " (_Float16, mangled)"
    return name.split("<", 1)[0]


def in_library(so):
    out = subprocess.run(["nm", "-C", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = set()
    for line in out.splitlines():
        parts = line.split(" ", 2)
        if len(parts) == 3 and "__device_stub__" in parts[2]:
            names.add(norm(parts[2]))
    return names


def main():
    argv = list(sys.argv[1:])
    json_out = None
    if "--json" in argv:
        i = argv.index("--json")
        json_out = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    so, logs = args[0], args[1:]
    have = in_library(so)
    raw = set()
    for p in logs:
        raw.update(l.strip() for l in open(p) if l.strip() and l.strip() != "?")
    used = {norm(n) for n in demangle(sorted(raw))}
    per = collections.OrderedDict()
    for n in sorted(have):
        per.setdefault(template_of(n), [0, 0])[0] += 1
    for n in used:
        per.setdefault(template_of(n), [0, 0])[1] += 1
    print("%-44s %8s %8s" % ("kernel template", "in .so", "selected"))
    for t, (a, b) in per.items():
        print("%-44s %8d %8d" % (t, a, b))
    unknown = sorted(used - have)
    print("%-44s %8d %8d   (picked kernels; launched directly: the rest)" % ("total", len(have), len(used)))
    if unknown:
        print("selected but not found among the host stubs (name normalisation?):", len(unknown))
        for n in unknown[:10]:
            print("   ", n)
    if "--unused" in sys.argv:
        for n in sorted(have - used):
            print("unused:", n)
    if json_out:
        json.dump({"in_library": sorted(have), "selected": sorted(used)}, open(json_out, "w"), indent=0)


if __name__ == "__main__":
    main()
