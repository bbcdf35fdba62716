#!/bin/bash
# same-box A/B of environment settings on the working tree: tools/ab_env.sh <tag> <rounds> "<VAR=val ... [-- bench args]>" ...
tag=$1; rounds=$2; shift 2
mkdir -p gpurun_out/r04
for i in $(seq $rounds); do
  k=0
  for setting in "$@"; do
    k=$((k+1))
    envpart="${setting%%--*}"; argpart=""; case "$setting" in *--*) argpart="${setting#*--}";; esac
    env $envpart python bench.py --no-cpu-baseline $argpart 2>/dev/null | grep '^{' >> gpurun_out/r04/${tag}_$k.jsonl
  done
done
k=0
for setting in "$@"; do k=$((k+1)); python - "$setting" gpurun_out/r04/${tag}_$k.jsonl <<'PY'
import json, sys
v = [json.loads(l) for l in open(sys.argv[2])]
print("%-60s" % sys.argv[1], " ".join("%.4f" % d["ms_per_step"] for d in v), "| trace in-frame", " ".join("%.4f" % d["roofline"]["kernel_ms"] for d in v))
PY
done
