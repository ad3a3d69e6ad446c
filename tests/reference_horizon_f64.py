#!/usr/bin/env python3
"""One-off check (a script, not collected by pytest; needs oracle/_ref/libnbody_ref_hip_f64.so): C5 - BASELINE.json configs[4],
N=1048576, fp64, radii 0 - over its WHOLE 100-step horizon, the fp64 product against the reference's own kernel text read
at double precision (oracle/ref_hip, REF_FLOAT_AS_DOUBLE) on the same GPU, state compared bit for bit every `chunk` steps.
The reference text takes about 3.2 s per step here: the suite runs two steps of it, this script all hundred (5.5 minutes).
    python tests/reference_horizon_f64.py [N] [steps] [chunk]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
bodies = nb.init_bodies(cfg, nb.F64)
st = nb.Stepper(cfg, precision=nb.F64)
st.upload(bodies)
st.set_kernel_timing(True)
blk = bodies.contiguousData.copy()
dt, gr = float(np.float32(cfg.timestep)), float(np.float32(cfg.growthRate))
cur, ref_ms, done, t0 = n, 0.0, 0, time.time()
while done < steps:
    k = min(chunk, steps - done)
    cur, ms, _ = ol.ref_hip_run(blk, cur, k, dt, cfg.fieldWidth, cfg.fieldHeight, gr)
    ref_ms += ms
    st.step(k)
    out = st.download()
    done += k
    same = out.numBodies == cur and np.array_equal(out.block.view(np.uint64), blk[:6 * cur].view(np.uint64))
    s = st.stats()
    print("step %4d: %d bodies, fp64 product %s reference text at double precision (bitwise); kernel time so far: reference %.1f s, "
          "product %.1f s (%.2fx); wall %.0f s" % (done, cur, "==" if same else "!=", ref_ms / 1e3, s.force_kernel_ms / 1e3,
                                                   ref_ms / max(s.force_kernel_ms, 1e-9), time.time() - t0), flush=True)
    if not same:
        sys.exit(1)
st.close()
print("OK: N=%d fp64, %d steps, product bit-identical to the reference's kernel text read at double precision throughout" % (n, steps))
