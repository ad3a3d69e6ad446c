#!/usr/bin/env python3
"""How the force kernels take collision-heavy states: N bodies with stock radii in ever smaller fields, the first `steps`
steps (force-kernel time per step, bodies left), ring kernel (automatic choice) against the one-lane kernel (variant 31).
    python3 dense_probe.py [N] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for field in (100000, 40000, 20000, 10000, 5000):
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field)
    bodies = nb.init_bodies(cfg)
    out = []
    for variant in (0, 31):
        st = nb.Stepper(cfg, kernel_variant=variant)
        st.upload(bodies)
        st.set_kernel_timing(True)
        ms = []
        last = 0.0
        for s in range(steps):
            st.step(1)
            t = st.stats().force_kernel_ms
            ms.append(t - last)
            last = t
        out.append((variant, ms, st.stats().n_bodies))
        st.close()
    print("N=%d field %6d: " % (n, field) + "   ".join("variant %2d: %s ms -> %d bodies" % (v, " ".join("%.2f" % m for m in ms), nn)
                                                        for v, ms, nn in out), flush=True)
