"""Aggregate two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each in its own run with --kernel-trace
only) into HBM bytes per launch per kernel.

usage: python tools/pmc_aggregate.py <fetch_dir> <write_dir> <out_prefix> [header text] [meta json]

`meta json` (e.g. '{"workload": "seg", "batch": 8, "height": 320, "width": 704, "conv_math": "bf16x3", "hn_version": 6,
"commit": "abc1234"}') is stored under "_meta" in the .json: bench.py only quotes the figures for a run that matches it.

FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half their
bytes); both counters are in KB.  Writes <out_prefix>.txt (table) and <out_prefix>.json (kernel -> figures).
"""
import csv
import glob
import json
import os
import re
import sys


def kernel_key(name):
    """conv_igemm_kernel<2, 2, 4, 32, 1> from the demangled signature (namespaces and argument list dropped)."""
    name = name.replace("(anonymous namespace)::", "").replace("hnconv::", "")
    name = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].strip()[:60]


def collect(directory, counter):
    per = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = kernel_key(row["Kernel_Name"])
                e = per.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    header = sys.argv[4] if len(sys.argv) > 4 else ""
    meta = json.loads(sys.argv[5]) if len(sys.argv) > 5 else None
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    rows = []
    for k, (calls, fsum) in fetch.items():
        wcalls, wsum = write.get(k, (0, 0.0))
        f_kb = fsum / max(calls, 1)
        w_kb = wsum / max(wcalls, 1)
        rows.append((k, calls, f_kb, w_kb, (2.0 * f_kb + w_kb) * 1024.0))
    rows.sort(key=lambda r: -r[1] * r[4])
    with open(out + ".txt", "w") as f:
        if header:
            f.write(header.rstrip() + "\n\n")
        f.write("%-46s %6s %22s %22s %s\n" % ("kernel", "calls", "FETCH_SIZE KB/launch", "WRITE_SIZE KB/launch",
                                             "HBM MB/launch = (2F+W)*1024/1e6"))
        for k, calls, f_kb, w_kb, b in rows[:40]:
            f.write("%-46s %6d %22.0f %22.0f %28.1f\n" % (k[:46], calls, f_kb, w_kb, b / 1e6))
    data = {k: {"calls": calls, "fetch_kb_per_launch": f_kb, "write_kb_per_launch": w_kb, "hbm_bytes_per_launch": b}
            for k, calls, f_kb, w_kb, b in rows}
    if meta is not None:
        data["_meta"] = meta
    with open(out + ".json", "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
