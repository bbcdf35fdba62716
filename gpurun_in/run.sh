for f in gpurun_in/lib_s*.so; do cp $f raytracedggx_amd/librtggx.so; timeout -k 10 120 python bench.py --steps 128 --warmup 32 --no-cpu-baseline > gpurun_out/sw.log 2>&1; python - $f <<'PY'
import json,sys
for l in open("gpurun_out/sw.log"):
    if l.startswith("{"):
        d=json.loads(l); p=d["passes_ms"]; print(sys.argv[1], "frame %.4f  kernel(ring) %.4f | serial trace %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], p["ray_trace_kernel"]))
PY
done
