"""From a rocprofv3 --kernel-trace CSV of a long bench run: per group of 16 frames the period (trace launch to trace launch), the mean duration of
every kernel of the frame, and where in the traversal's period the other stages' kernels START (their phase, in % of the period): what differs
between the pipeline's two states (profiles/r04_k_states.txt).   python tools/state_trace.py <kernel_trace.csv> [first_group] [groups]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace("void ", "").replace("rt::", "").split("(")[0]
    return {"spatialTiledKernel<0>": "H", "spatialTiledKernel<1>": "V", "temporalKernel": "temporal", "toneMapKernel": "tone", "rasterSmall": "rSmall", "rasterLarge": "rLarge",
            "rayGenKernel": "rayGen", "shadeKernel": "shade"}.get(n, "trace" if n.startswith("traceKernel") else n)
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
names = ["rSmall", "rLarge", "rayGen", "trace", "shade", "H", "V", "temporal", "tone"]
tr = [k for k in ks if k[2] == "trace"]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
print("group  period | " + " ".join("%8s" % n for n in names) + " | start phase (%% of the period after the trace launch): " + " ".join(n for n in names if n != "trace"))
g = 0
for i in range(first * 16, len(tr) - 16, 16):
    if g >= groups: break
    t0, t1 = tr[i][0], tr[i + 16][0]
    period = (t1 - t0) / 16e3
    if period > 400: g += 1; continue
    dur = {n: [] for n in names}; phase = {n: [] for n in names}
    starts = [t[0] for t in tr[i:i + 17]]
    for s, e, n in ks:
        if t0 <= s < t1 and n in dur:
            dur[n].append((e - s) / 1e3)
            j = max(k for k in range(17) if starts[k] <= s)
            phase[n].append((s - starts[j]) / 1e3 / period * 100.0)
    print("%5d  %6.1f | " % (i // 16, period) + " ".join("%8.1f" % (sum(dur[n]) / max(len(dur[n]), 1)) for n in names) + " | " + " ".join("%5.0f" % (sum(phase[n]) / max(len(phase[n]), 1)) for n in names if n != "trace"))
    g += 1
