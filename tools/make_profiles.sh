#!/bin/bash
# Regenerates the judged summaries under profiles/ for one tag (run on the GPU box through gpurun, from the repo root):
#   tools/make_profiles.sh r01_h
# 1. counter passes (tools/pmc.sh groups 1,2,10,11)  -> profiles/<tag>_pmc_report.txt, profiles/<tag>_pmc_traffic.json
# 2. the default bench line                          -> profiles/<tag>_bench.json
# 3. rocprofv3 --kernel-trace --stats of the same    -> profiles/<tag>_kernel_stats.csv
# Raw rocprofv3 output goes to gpurun_out/profiles_<tag>/ (scratch).  On the GPU box only gpurun_out/ travels back: run
#   tools/make_profiles.sh <tag> --summaries-only
# afterwards in the repo to rebuild profiles/<tag>_* from the raw files that came back.
set -e
tag=${1:?tag}
export TMPDIR=/tmp
out=gpurun_out/profiles_$tag
mkdir -p "$out" profiles
summaries_only=$2
if [ "$summaries_only" != "--summaries-only" ]; then
tools/pmc.sh "$out/pmc" 1 2 > /dev/null
tools/pmc.sh "$out/pmc" 4 4 > /dev/null      # TCP_TOTAL_CACHE_ACCESSES: the vector L1's request rate (roofline.l1_requests)
tools/pmc.sh "$out/pmc" 9 11 > /dev/null     # GRBM_GUI_ACTIVE (cycles), FETCH_SIZE, WRITE_SIZE
fi
# the counter summaries come first: bench.py quotes the newest profiles/*_pmc_traffic.json in its roofline.traffic
python3 tools/pmc_report.py "$out/pmc" > "profiles/${tag}_pmc_report.txt"
python3 - "$tag" <<'EOF'
import json, re, sys
tag = sys.argv[1]
txt = open("profiles/%s_pmc_report.txt" % tag).read()
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `python3 bench.py --steps 4 --warmup 2` (tools/pmc.sh groups 10, 11), per dispatch averages",
       "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE is in KB and counts 128-byte requests as 64 on gfx950 (MI355X_MICROARCH.md 'HBM'); uncalibrated for narrow gathers",
       "workload": "bunny 1920x1080, all-metal, shared-memory denoiser", "kernels": {}}
# the trace kernel is a template (waves per workgroup, waves per SIMD, table in LDS): the variant with the most dispatches is "the" kernel
def dispatches(block):
    m = re.search(r"\((\d+) dispatches\)", block)
    return int(m.group(1)) if m else 0
blocks = re.split(r"\n(?=\S)", txt)
variants = [b for b in blocks if b.split("\n")[0].strip().startswith("void rt::traceKernel<")]
main_variant = max(variants, key=dispatches).split("\n")[0].strip() if variants else None
def kernel_name(block):
    name = block.split("\n")[0].strip()
    return "rt::traceKernel" if name == main_variant else name
out["trace_kernel_variant"] = main_variant
for block in blocks:
    name = kernel_name(block)
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([0-9.]+) per dispatch", block, re.M)}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and name.startswith(("rt::", "void rt::")):
        out["kernels"][name] = {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                                "traffic_bytes": int(1024 * (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]))}
# wave-level VALU instructions of the frame's kernels (SQ_INSTS_VALU per dispatch, one dispatch of each per frame): bench.py reports
# the share of the VALU issue slots they take (2 cycles per wave64 instruction on a SIMD-32, 1024 SIMDs, 2.4 GHz; profiles/r02_a_valu_issue.txt)
frame_kernels = ("rt::clearVisDepth", "rt::rasterSmall", "rt::rasterLarge", "rt::rayGenKernel", "rt::traceKernel", "rt::shadeKernel",
                 "void rt::spatialTiledKernel<0>", "void rt::spatialTiledKernel<1>", "void rt::spatialTiledKernel<2>", "void rt::spatialTiledKernel<3>",
                 "rt::temporalKernel", "rt::toneMapKernel")
valu = 0.0
for block in blocks:
    name = kernel_name(block)
    m = re.search(r"^\s+SQ_INSTS_VALU\s+([0-9.]+) per dispatch", block, re.M)
    if m and name.split("(")[0] in frame_kernels:
        valu += float(m.group(1))
out["valu_instructions_per_frame"] = int(valu)
# requests the trace kernel puts to the vector L1 per CU and cycle (DESIGN.md "what bounds it": a traversal step costs its divergent lane
# requests, and the L1 takes ~one per cycle per CU): TCP_TOTAL_CACHE_ACCESSES / (CUs x GRBM_GUI_ACTIVE), both per dispatch of the
# (serialised) counter passes
for block in blocks:
    if kernel_name(block) == "rt::traceKernel":
        vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([0-9.]+) per dispatch", block, re.M)}
        if "TCP_TOTAL_CACHE_ACCESSES" in vals and vals.get("GRBM_GUI_ACTIVE", 0) > 0:
            cycles = vals["GRBM_GUI_ACTIVE"] / 8.0      # the counter is summed over the 8 XCDs (2.66 M for a 0.138 ms launch at 2.4 GHz)
            out["trace_kernel_l1"] = {"requests_per_launch": int(vals["TCP_TOTAL_CACHE_ACCESSES"]), "cycles_per_launch": int(cycles), "cus": 256,
                                      "requests_per_cu_cycle": round(vals["TCP_TOTAL_CACHE_ACCESSES"] / (256.0 * cycles), 4),
                                      "source": "rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES and --pmc GRBM_GUI_ACTIVE / 8 XCDs (tools/pmc.sh groups 4, 9); the launch alone on the chip (counter passes serialise the kernels)"}
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print("trace kernel traffic:", out["kernels"].get("rt::traceKernel"))
EOF
if [ "$summaries_only" != "--summaries-only" ]; then
python3 bench.py > "$out/bench.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace" -o run --output-format csv -- python3 bench.py --no-cpu-baseline > "$out/trace.log" 2>&1
fi
grep '^{' "$out/bench.log" | tail -n 1 > "profiles/${tag}_bench.json"
cp "$out/trace/run_kernel_stats.csv" "profiles/${tag}_kernel_stats.csv"
echo "profiles/${tag}_* written"
