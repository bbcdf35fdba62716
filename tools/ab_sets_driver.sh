#!/bin/bash
# input sets 4 / 5 at the driver's settings (--steps 20 --warmup 5), eight runs per build, alternating builds twice
A="capi visibility lbvh raytrace trace denoise env"
for n in "$@"; do
  for o in $A; do rm -f raytracedggx_amd/_build/$o.o; done
  make -C raytracedggx_amd EXTRA="-DRT_SETS=$n" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  line="sets $n:"
  for i in 1 2 3 4 5 6 7 8; do v=$(python bench.py --no-cpu-baseline --sustained-frames 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])"); line="$line $v"; done
  echo "$line"
done
for o in $A; do rm -f raytracedggx_amd/_build/$o.o; done; make -C raytracedggx_amd > /dev/null 2>&1
