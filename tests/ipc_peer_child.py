"""Child process of test_history_images_of_another_process_through_hip_ipc (tests/test_gpu_parity.py): renders whole frames with the
product, exports its two history images as hipIpc handles, and after every frame leaves the temporal result and the back buffer in
<dir>/tss_<f>.npy / bb_<f>.npy for the parent to compare its strip with.  Protocol on stdin / stdout: "frame" -> "done <f>", "quit"."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracedggx_amd import app, capi  # noqa: E402

out = sys.argv[1]
a = app.RayTracedGGX(sys.argv[2:])
print("handles " + a.context.history_ipc_export().hex(), flush=True)
f = 0
for line in sys.stdin:
    if line.strip() == "quit":
        break
    a.OnUpdate(); a.OnRender(); a.context.sync()
    np.save(os.path.join(out, "tss_%d.npy" % f), a.context.readback(capi.BUF_TSS0 + a.context.frame_parity()))
    np.save(os.path.join(out, "bb_%d.npy" % f), a.context.readback(capi.BUF_BACKBUFFER))
    print("done %d" % f, flush=True)
    f += 1
a.OnDestroy()
