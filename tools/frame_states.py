"""Two steady states of the frame pipeline, side by side, from a rocprofv3 --kernel-trace CSV of a free-running run: frames (one per
traversal launch) are classed by the mean period of the 16 frames around them (below / above the run's median + 4 %), and for each class
the mean start and end of every kernel of the frame relative to its traversal's start are printed, with the queue they ran on.
   python tools/frame_states.py run_kernel_trace.csv [first] [count]"""
import csv, sys, bisect, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 300
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
def short(n): return n.replace("void ", "").replace("rt::", "").split("(")[0]
by = {}
for r in rows:
    by.setdefault(short(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])))
for v in by.values(): v.sort()
tr = [k for name, v in by.items() if name.startswith("traceKernel") for k in v]; tr.sort()
last = min(first + count, len(tr) - 20)
period = [(tr[k + 8][0] - tr[k - 8][0]) / 16e3 for k in range(first, last)]
med = statistics.median(period)
print("frames %d..%d, median period %.1f us; fast: below %.1f, slow: above" % (first, last, med, med * 1.04))
# a frame's kernels: per name, the launch whose start is nearest to the expected offset -- take launches in order instead: kernel n of name X belongs to frame n (one launch per frame)
names = [n for n, v in by.items() if abs(len(v) - len(tr)) <= 8 and not n.startswith("traceKernel")]
def offs(cls):
    acc = {n: [0.0, 0.0, 0, None] for n in names}; acc["traceKernel"] = [0.0, 0.0, 0, None]; n_frames = 0; per = 0.0
    for i, k in enumerate(range(first, last)):
        slow = period[i] > med * 1.04
        if slow != cls: continue
        n_frames += 1; per += period[i]
        s = tr[k][0]
        acc["traceKernel"][1] += (tr[k][1] - s) / 1e3; acc["traceKernel"][2] += 1; acc["traceKernel"][3] = tr[k][2]
        for n in names:
            v = by[n]; d = len(v) - len(tr)      # launches before the first traversal (set-up) shift the index
            j = k + d
            # the launch of this frame: the one with the same index counted from the END (every frame launches each kernel once)
            if 0 <= j < len(v):
                a = acc[n]; a[0] += (v[j][0] - s) / 1e3; a[1] += (v[j][1] - s) / 1e3; a[2] += 1; a[3] = v[j][2]
    return n_frames, per / max(n_frames, 1), {n: (a[0] / a[2], a[1] / a[2], a[3]) for n, a in acc.items() if a[2]}
for cls, label in ((False, "FAST"), (True, "SLOW")):
    n, p, o = offs(cls)
    print("%s state: %d frames, mean period %.1f us; kernel: start .. end relative to the frame's traversal start (us), queue" % (label, n, p))
    for name, (a, b, q) in sorted(o.items(), key=lambda kv: kv[1][0]):
        print("   %-28s %8.1f .. %8.1f   (%.1f)  q%s" % (name, a, b, b - a, q))
