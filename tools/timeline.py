"""Prints the kernel timeline of a few steady-state frames from a rocprofv3 --kernel-trace CSV (run_kernel_trace.csv):
python tools/timeline.py <csv> [first trace launch] [launches]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
def short(n):
    return n.replace("void ", "").replace("rt::", "").split("(")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])) for r in rows)
tr = [k for k in ks if k[3].startswith("traceKernel")]
t0, t1 = tr[first][0] - 150000, tr[first + count][1]
last_end = {}
for s, e, q, n in ks:
    if t0 <= s <= t1:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        print("q%d %-26s start %8.1f  end %8.1f  dur %6.1f  gap on its queue %5.1f" % (q, n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
    last_end[q] = e
