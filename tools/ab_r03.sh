#!/bin/bash
# same-box A/B of trees built in the container under _ab/ against the working tree:  tools/ab_r03.sh <tag> <rounds> <tree>...   ("." = the working tree)
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out/r04
for i in $(seq $rounds); do
  for t in "$@"; do
    name=$(echo $t | tr '/. ' '___')
    if [ "$t" = "." ]; then python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/r04/${tag}_new.jsonl
    elif [ "$t" = ".fused" ]; then python bench.py --no-cpu-baseline --tone-map fused 2>/dev/null | grep "^{" >> gpurun_out/r04/${tag}_new_fused.jsonl
    else (cd _ab/$t && python bench.py --no-cpu-baseline 2>/dev/null | grep '^{') >> gpurun_out/r04/${tag}_$name.jsonl; fi
  done
done
python - $tag <<'PY'
import json, glob, sys
for f in sorted(glob.glob("gpurun_out/r04/%s_*.jsonl" % sys.argv[1])):
    v = [json.loads(l) for l in open(f)]
    print("%-40s" % f.split("/")[-1], " ".join("%.4f" % d["ms_per_step"] for d in v), "| trace in-frame", " ".join("%.4f" % d["roofline"]["kernel_ms"] for d in v), "| waves", " ".join(str(d["config"].get("trace_workgroup_waves")) for d in v))
PY
