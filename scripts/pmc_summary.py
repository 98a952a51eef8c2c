"""Summarise rocprofv3 --pmc csv output (one directory per pass) per kernel: mean counter value per dispatch.

    python scripts/pmc_summary.py gpurun_out/pmc22 > profiles/rNN_pmc_k22.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                k = k.replace("void g16::(anonymous namespace)::", "").replace("g16::", "")
                k = k.split("(", 2)[0] if not k.startswith("(") else k
                k = k.replace("(anonymous namespace)::", "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        print(k)
        for c, v in sorted(ctrs.items()):
            print(f"   {c:34s} n={len(v):3d} mean={sum(v) / len(v):18.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
