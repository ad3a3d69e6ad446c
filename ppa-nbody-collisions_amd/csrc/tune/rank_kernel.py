#!/usr/bin/env python3
"""Profiling probe: ONE rank's force kernel of a G-rank partition of N bodies, launched `reps` times back to back
on one GPU in steady state (nbody_debug_force_only).  Also the target of the rocprofv3 counter passes.
    python3 rank_kernel.py N G rank variant reps [stock|r0] [fp64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n, world, rank, variant, reps = (int(v) for v in sys.argv[1:6])
kw = {} if (len(sys.argv) > 6 and sys.argv[6] == "stock") else {"minRadius": 0.0, "maxRadius": 0.0}
cfg = nb.stock_config(particleCount=n, **kw)
precision = nb.F64 if "fp64" in sys.argv[6:] else nb.F32
st = nb.Stepper(cfg, precision=precision, rank=rank, world=world, group=world > 1, kernel_variant=variant)
st.upload(nb.init_bodies(cfg, precision))
st.force_only(3)
st.sync()
s0 = st.stats()
st.set_kernel_timing(True)
st.force_only(reps)
s1 = st.stats()
ms = s1.force_kernel_ms / s1.force_kernel_launches
pairs = (s1.pairs - s0.pairs) / reps
print("N=%d rank %d of %d (own %d) variant %d: %.3f ms per force launch, %.3e pairs per launch -> %.3e pairs/s "
      "on this rank, x%d = %.3e" % (n, rank, world, s1.n_own, variant, ms, pairs, pairs / ms * 1e3, world,
                                   world * pairs / ms * 1e3), flush=True)
st.close()
