#!/bin/bash
# average in-frame duration per kernel of a bench run, through rocprofv3 --kernel-trace --stats:  tools/kstats.sh <tag> <tree|.> [bench args]
tag=$1; tree=$2; shift 2
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r04/ks_$tag; mkdir -p $out
dir=$root; [ "$tree" != "." ] && dir=$root/_ab/$tree
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python3 $dir/bench.py --steps 256 --warmup 64 --no-cpu-baseline "$@" > $out/bench.log 2>&1
cd $root
python - $out/ks_kernel_stats.csv $tag <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].replace("void ", "").replace("rt::", "").split("(")[0]
    if int(r["Calls"]) > 200 and any(k in n for k in ("raster", "rayGen", "trace", "shade", "spatial", "temporal", "toneMap")):
        print("%-10s %-28s calls %5s  avg %7.1f us" % (sys.argv[2], n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep '^{' $out/bench.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$tag', 'ms_per_step', d['ms_per_step'])"
find $out -name "*.csv" -size +1M -delete
