"""Sum the counter CSVs written by tools/pmc.sh per kernel and print per-dispatch averages.
   python tools/pmc_report.py gpurun_out/pmc [kernel-substring ...]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum over dispatches
disp = defaultdict(lambda: defaultdict(set))       # kernel -> counter -> dispatch ids
for path in sorted(glob.glob(root + "/g*/**/*counter_collection.csv", recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            if want and not any(w in k for w in want):
                continue
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            disp[k][c].add(row["Dispatch_Id"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = max(len(disp[k][c]), 1)
        print("   %-36s %16.1f per dispatch  (%d dispatches)" % (c, acc[k][c] / n, n))
