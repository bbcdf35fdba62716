"""Per-frame phases from a rocprofv3 --kernel-trace CSV of a free-running run: for every traversal launch k (one per frame) the period since
the previous one, its duration, and when the other kernels of frame k ran relative to it -- to see which stage the frame waits for, and
whether a run has more than one steady state.   python tools/frame_phases.py run_kernel_trace.csv [first] [count]"""
import csv, sys, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
def short(n): return n.replace("void ", "").replace("rt::", "").split("(")[0].split("<")[0]
by = {}
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for v in by.values(): v.sort()
tr = by["traceKernel"]
def last_before(name, t):      # the launch of `name` that ended last before t
    v = by.get(name, []); i = bisect.bisect_right([e for s, e in v], t) - 1
    return v[i] if i >= 0 else None
def first_after(name, t):
    v = by.get(name, []); i = bisect.bisect_left([s for s, e in v], t)
    return v[i] if i < len(v) else None
print("frame  period  trace  | rayGen end -> trace start | trace end -> shade start | shade start -> temporal end | refitTris start -> rasterSmall start (frame's own) | rasterSmall dur rayGen dur")
out = []
for k in range(first, min(first + count, len(tr) - 1)):
    s, e = tr[k]
    rg = last_before("rayGenKernel", s); sh = first_after("shadeKernel", e)
    tp = first_after("temporalKernel", sh[0]) if sh else None
    rs = last_before("rasterSmall", rg[0]) if rg else None
    rf = last_before("refitTris", rs[0]) if rs else None
    out.append(((s - tr[k - 1][0]) / 1e3, (e - s) / 1e3, (s - rg[1]) / 1e3 if rg else -1, (sh[0] - e) / 1e3 if sh else -1, (tp[1] - sh[0]) / 1e3 if tp else -1,
                (rs[0] - rf[0]) / 1e3 if rf else -1, (rs[1] - rs[0]) / 1e3 if rs else -1, (rg[1] - rg[0]) / 1e3 if rg else -1))
for i in range(0, len(out), 16):      # means of 16 frames
    blk = out[i:i + 16]
    print("%5d " % (first + i) + "  ".join("%7.1f" % (sum(b[j] for b in blk) / len(blk)) for j in range(8)))
