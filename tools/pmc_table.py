"""Per-kernel mean of every counter found under a directory of rocprofv3 --pmc runs.
usage: python tools/pmc_table.py <dir> [kernel-substring]"""
import csv
import glob
import os
import sys

from pmc_aggregate import kernel_key


def main():
    root, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = kernel_key(row["Kernel_Name"])
                if want not in k:
                    continue
                e = acc.setdefault((k, row["Counter_Name"]), [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    for (k, c), (n, v) in sorted(acc.items()):
        print("%-44s %-28s launches %5d  mean %16.1f" % (k[:44], c, n, v / n))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
