#!/usr/bin/env python3
"""Development probe for profilers: steps a G-rank partition of N bodies on ONE GPU (ranks serialised) with
one kernel variant.   python3 rank_probe.py N G variant [steps]"""
import os
import sys

os.environ["NBODY_GROUP_SERIALIZE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n, world, variant = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
grp = nb.StepperGroup(world, cfg=cfg, kernel_variant=variant)
grp.upload(nb.init_bodies(cfg))
for r in grp.ranks:
    r.set_kernel_timing(True)
grp.step(steps)
s = grp.ranks[0].stats()
print("N=%d G=%d variant=%d: %.3f ms per force launch (rank 0)" % (n, world, variant, s.force_kernel_ms / s.force_kernel_launches))
grp.close()
