for f in gpurun_in/lib_a_default.so gpurun_in/lib_b_noslp_rt_dn.so gpurun_in/lib_a_default.so gpurun_in/lib_b_noslp_rt_dn.so gpurun_in/lib_a_default.so gpurun_in/lib_b_noslp_rt_dn.so; do cp $f raytracedggx_amd/librtggx.so; timeout -k 10 120 python bench.py --steps 128 --warmup 32 --no-cpu-baseline > gpurun_out/sw.log 2>&1; python - $f <<'PY'
import json,sys
for l in open("gpurun_out/sw.log"):
    if l.startswith("{"):
        d=json.loads(l); p=d["passes_ms"]; print(sys.argv[1], "frame %.4f  kernel(ring) %.4f | serial: vis %.4f rt %.4f trace %.4f H %.4f V %.4f T %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], p["visibility"], p["ray_trace"], p["ray_trace_kernel"], p["spatial_refl_h"], p["spatial_refl_v"], p["temporal"]))
PY
done
