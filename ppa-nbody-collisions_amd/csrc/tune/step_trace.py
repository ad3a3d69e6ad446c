#!/usr/bin/env python3
"""Development probe: per-step, per-rank force-kernel time of a G-rank partition on ONE GPU (ranks serialised).
    python3 step_trace.py N G variant steps [stock]"""
import os
import sys

os.environ["NBODY_GROUP_SERIALIZE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n, world, variant, steps = (int(v) for v in sys.argv[1:5])
kw = {} if (len(sys.argv) > 5 and sys.argv[5] == "stock") else {"minRadius": 0.0, "maxRadius": 0.0}
cfg = nb.stock_config(particleCount=n, **kw)
grp = nb.StepperGroup(world, cfg=cfg, kernel_variant=variant)
grp.upload(nb.init_bodies(cfg))
for r in grp.ranks:
    r.set_kernel_timing(True)
prev = [0.0] * world
for s in range(steps):
    grp.step(1)
    st = [r.stats() for r in grp.ranks]
    ms = [x.force_kernel_ms for x in st]
    out = grp.ranks[0].download()
    big = int((np.abs(out.Positions) >= 2.0 ** 38).any(axis=1).sum() + np.isnan(out.Positions).any(axis=1).sum())
    print("step %2d  N=%d  unbounded bodies %d  own %s  ms/rank: %s" %
          (s, st[0].n_bodies, big, " ".join(str(x.n_own) for x in st),
           " ".join("%.3f" % (a - b) for a, b in zip(ms, prev))), flush=True)
    prev = ms
grp.close()
