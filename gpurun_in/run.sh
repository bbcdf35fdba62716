for f in gpurun_in/lib_v_*.so; do cp $f raytracedggx_amd/librtggx.so; timeout -k 10 120 python bench.py --steps 64 --warmup 16 --no-cpu-baseline > gpurun_out/sw.log 2>&1; python - $f <<'PY'
import json,sys
for l in open("gpurun_out/sw.log"):
    if l.startswith("{"):
        d=json.loads(l); p=d["passes_ms"]; print(sys.argv[1], "frame %.4f  refl_v %.4f diff_v %.4f refl_h %.4f" % (d["ms_per_step"], p["spatial_refl_v"], p["spatial_diff_v"], p["spatial_refl_h"]))
PY
done
