#!/bin/bash
# input sets (frames in flight) 4 / 5 / 6: the contract's window and the 1024-frame one behind it (bench.py `sustained`), three runs per build
A="capi visibility lbvh raytrace trace denoise env"
mkdir -p gpurun_out/r04
for n in "$@"; do
  for o in $A; do rm -f raytracedggx_amd/_build/$o.o; done
  make -C raytracedggx_amd EXTRA="-DRT_SETS=$n" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for i in 1 2 3; do python bench.py --no-cpu-baseline --sustained-frames 2048 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sets $n: window %.4f  sustained(2048) %.4f  trace in frame %.4f' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['kernel_ms']))"; done
done
for o in $A; do rm -f raytracedggx_amd/_build/$o.o; done; make -C raytracedggx_amd > /dev/null 2>&1
