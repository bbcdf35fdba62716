"""Repeats the 8-strip acceptance run of tests/test_gpu_parity.py (test_strips_equal_the_single_context_at_any_velocity) N times in ONE process
and counts the runs in which a strip differs from the single context: a hunt for an intermittent difference (1 in ~7 full-suite runs).
python tools/probes/strips_repeat.py N"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for i in range(n):
    try:
        T._strips_through_rccl_equal_the_full_frame(1280, 720, 8, True, 60, extra=("-metallic", 0.25, 0.5), overreach=[])
        print("run %d ok" % i, flush=True)
    except AssertionError as e:
        bad += 1
        print("run %d FAILED: %s" % (i, str(e).strip().splitlines()[:6]), flush=True)
print("%d of %d runs differed" % (bad, n))
