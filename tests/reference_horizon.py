#!/usr/bin/env python3
"""One-off check (a script, not collected by pytest; needs oracle/_ref/libnbody_ref_hip.so): a BASELINE configuration over
its WHOLE horizon, the product against the reference's own kernels on the same GPU, state compared bit for bit every
`chunk` steps.  C4 (N=262144, 1000 steps) takes the reference's kernels about 3.5 minutes.
    python tests/reference_horizon.py [N] [steps] [chunk] [stock|r0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import ppa_nbody_collisions_amd as nb  # noqa: E402
import oracle_lib as ol  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kw = {} if (len(sys.argv) > 4 and sys.argv[4] == "stock") else {"minRadius": 0.0, "maxRadius": 0.0}
cfg = nb.stock_config(particleCount=n, **kw)
bodies = nb.init_bodies(cfg)
st = nb.Stepper(cfg)
st.upload(bodies)
st.set_kernel_timing(True)
blk = bodies.contiguousData.copy()
cur, ref_ms, done, t0 = n, 0.0, 0, time.time()
while done < steps:
    k = min(chunk, steps - done)
    cur, ms, _ = ol.ref_hip_run(blk, cur, k, np.float32(cfg.timestep), cfg.fieldWidth, cfg.fieldHeight,
                                np.float32(cfg.growthRate))
    ref_ms += ms
    st.step(k)
    out = st.download()
    done += k
    same = out.numBodies == cur and np.array_equal(out.block.view(np.uint32), blk[:6 * cur].view(np.uint32))
    s = st.stats()
    print("step %4d: %d bodies, product %s reference kernels (bitwise); kernel time so far: reference %.1f s, product %.2f s "
          "(%.1fx); wall %.0f s" % (done, cur, "==" if same else "!=", ref_ms / 1e3, s.force_kernel_ms / 1e3,
                                    ref_ms / max(s.force_kernel_ms, 1e-9), time.time() - t0), flush=True)
    if not same:
        sys.exit(1)
st.close()
print("OK: N=%d, %d steps, product bit-identical to the reference's own kernels throughout" % (n, steps))
