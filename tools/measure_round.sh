#!/bin/bash
# The judged measurement set of a round, on the GPU box (through gpurun, from the repo root):  tools/measure_round.sh r04_p
#   profiles/<tag>_{pmc_report.txt,pmc_traffic.json,bench.json,kernel_stats.csv}   (tools/make_profiles.sh)
#   profiles/<tag>_bench_driver_settings.jsonl   ten runs of the driver's command: python bench.py --steps 20 --warmup 5
#   profiles/<tag>_other_configs.jsonl           DESIGN.md section 9's table
# Only gpurun_out/ travels back from the box: the script works there and `tools/measure_round.sh <tag> --collect` copies to profiles/ afterwards.
tag=${1:?tag}; out=gpurun_out/measure_$tag
if [ "$2" = "--collect" ]; then
  tools/make_profiles.sh $tag --summaries-only
  cp $out/bench_driver_settings.jsonl profiles/${tag}_bench_driver_settings.jsonl
  cp $out/other_configs.jsonl profiles/${tag}_other_configs.jsonl
  exit 0
fi
mkdir -p $out
tools/make_profiles.sh $tag > $out/make_profiles.log 2>&1; echo "profiles done"
for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' >> $out/bench_driver_settings.jsonl; done; echo "driver settings done"
o="--steps 128 --warmup 48 --no-cpu-baseline"
{ python bench.py $o --mesh dragon.obj; python bench.py $o --width 3840 --height 2160; python bench.py $o --mesh dragon.obj --width 3840 --height 2160
  python bench.py $o --metallic 0.25 0.5; python bench.py $o --mesh dragon.obj --metallic 0.25 0.5; python bench.py $o --width 1920 --height 171 --steps 512
  python bench.py $o --width 256 --height 144 --steps 512; python bench.py $o --deform 0.3; python bench.py $o --mesh dragon.obj --width 3840 --height 2160 --deform 0.3; } 2>/dev/null | grep '^{' > $out/other_configs.jsonl
echo "other configs done"
python - $out <<'PY'
import json, sys
v = [json.loads(l) for l in open(sys.argv[1] + "/bench_driver_settings.jsonl")]
ms = [d["ms_per_step"] for d in v]
print("driver settings: %s  mean %.4f  -%.1f%% / +%.1f%%" % (" ".join("%.4f" % x for x in ms), sum(ms) / len(ms), 100 * (1 - min(ms) / (sum(ms) / len(ms))), 100 * (max(ms) / (sum(ms) / len(ms)) - 1)))
for l in open(sys.argv[1] + "/other_configs.jsonl"):
    d = json.loads(l); print("%.4f ms  %.0f Mrays/s  %s" % (d["ms_per_step"], d["value"], d["config"]["workload"][12:110]))
PY
