#!/bin/bash
# A/B of compile-time variants on the GPU box (through gpurun, from the repo root):  tools/ab_build.sh <tag> "<objects>" "<EXTRA flags>" [bench args]
# rebuilds the named objects of librtggx.so with the flags, runs the bench line twice (256 steps) and appends the lines to
# gpurun_out/r04/ab_<tag>.jsonl; the default build is restored by the next call with empty flags.
tag=$1; objs=$2; extra=$3; shift 3
mkdir -p gpurun_out/r04
for o in $objs; do rm -f raytracedggx_amd/_build/$o.o; done
make -C raytracedggx_amd EXTRA="$extra" > /dev/null 2>&1 || { echo "build of $tag failed"; exit 1; }
for i in 1 2; do python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> gpurun_out/r04/ab_$tag.jsonl; done
python - "$tag" <<'PY'
import json, sys
for l in open("gpurun_out/r04/ab_%s.jsonl" % sys.argv[1]):
    d = json.loads(l); print(sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"])
PY
