timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -n 3 gpurun_out/t.log
for e in RTGGX_FIRST_PASS_STEPS=100000 RTGGX_FIRST_PASS_STEPS=32 RTGGX_FIRST_PASS_STEPS=24 RTGGX_FIRST_PASS_STEPS=16 RTGGX_FIRST_PASS_STEPS=12 RTGGX_FIRST_PASS_STEPS=8; do env $e timeout -k 10 120 python bench.py --steps 128 --warmup 32 --no-cpu-baseline > gpurun_out/sw.log 2>&1; python - $e <<'PY'
import json,sys
for l in open("gpurun_out/sw.log"):
    if l.startswith("{"):
        d=json.loads(l); p=d["passes_ms"]; print(sys.argv[1], "frame %.4f  kernel(ring) %.4f | serial: vis %.4f rt %.4f trace %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], p["visibility"], p["ray_trace"], p["ray_trace_kernel"]))
PY
done
