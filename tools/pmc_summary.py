#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs written by tools/pmc.sh: per kernel, the mean of every counter
over its dispatches (dispatches of the same kernel name and grid are pooled)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                key = (k[:90], row.get("Grid_Size", "?"))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key in sorted(acc):
        n = max(len(v) for v in acc[key].values())
        print(f"== {key[0]}  grid={key[1]}  dispatches={n}")
        for c in sorted(acc[key]):
            v = acc[key][c]
            print(f"   {c:45s} mean {sum(v) / len(v):.6g}   (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
