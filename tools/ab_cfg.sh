#!/bin/bash
# same-box A/B of one bench configuration between trees under _ab/ and the working tree:  tools/ab_cfg.sh <tag> <rounds> "<bench args>" <tree>...
tag=$1; rounds=$2; args=$3; shift 3
mkdir -p gpurun_out/r04
for i in $(seq $rounds); do
  for t in "$@"; do
    name=$(echo $t | tr '/. ' '___')
    if [ "$t" = "." ]; then python bench.py --no-cpu-baseline $args 2>/dev/null | grep "^{" >> gpurun_out/r04/${tag}_new.jsonl
    else (cd _ab/$t && python bench.py --no-cpu-baseline $args 2>/dev/null | grep '^{') >> gpurun_out/r04/${tag}_$name.jsonl; fi
  done
done
python - $tag "$args" <<'PY'
import json, glob, sys
for f in sorted(glob.glob("gpurun_out/r04/%s_*.jsonl" % sys.argv[1])):
    v = [json.loads(l) for l in open(f)]
    print("%-28s %-40s" % (f.split("/")[-1], sys.argv[2]), " ".join("%.4f" % d["ms_per_step"] for d in v))
PY
