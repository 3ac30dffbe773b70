"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter
over the dispatches in the file.  usage: python profiles/summarize_pmc.py <dir-with-pass-subdirs>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ".").replace(",", ".")
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    print("kernel," + ",".join(counters) + ",dispatches")
    for name in sorted(acc):
        if not name.startswith("offk"):
            continue
        vals = acc[name]
        n = max(len(v) for v in vals.values())
        print(name + "," + ",".join("%.6g" % (sum(vals[c]) / len(vals[c])) if c in vals else "" for c in counters) + ",%d" % n)


if __name__ == "__main__":
    main(sys.argv[1])
