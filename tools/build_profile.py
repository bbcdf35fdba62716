"""GPU time of rtggx_build_as from a rocprofv3 --kernel-trace CSV: per build (ground slab, model) the launches, the time from the first
kernel's start to the last one's end, the sum of the kernel durations, and the kernels by share.
   rocprofv3 --kernel-trace -d out -o run --output-format csv -- python3 bench.py --steps 4 --warmup 2 --prime-frames 8 --no-cpu-baseline
   python tools/build_profile.py out/**/run_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = ("buildBegin", "boundsKernel", "mortonKernel", "radixHist", "scanChunks", "scanTotals", "scanExclusive", "radixScatter", "plocInit", "plocNearest", "plocCount", "plocScatter",
         "plocFinal", "treeletRootsKernel", "buildTreeletsKernel", "planTopKernel", "depthKernel", "treeCostKernel", "emitTris", "emitNodes", "emitNodes4", "emitTop")
b = [r for r in rows if any(n in r["Kernel_Name"] for n in names)]
b.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(b) if "buildBegin" in r["Kernel_Name"]] + [len(b)]
for k in range(len(starts) - 1):
    seg = b[starts[k]:starts[k + 1]]
    # a build ends with its emit kernels; what follows (tree costs of later refits) belongs to no build
    last = max(i for i, r in enumerate(seg) if "emit" in r["Kernel_Name"] or "treeCost" in r["Kernel_Name"] or "depthKernel" in r["Kernel_Name"])
    seg = seg[:last + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("build %d: %d launches, first start to last end %.1f us, sum of kernel durations %.1f us" % (k, len(seg), (t1 - t0) / 1e3, busy / 1e3))
    agg = {}
    for r in seg:
        n = r["Kernel_Name"].split("(")[0].replace("rt::", "").replace("void ", "")
        a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for n, (cnt, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("   %-24s x%-3d %8.1f us" % (n, cnt, d / 1e3))
