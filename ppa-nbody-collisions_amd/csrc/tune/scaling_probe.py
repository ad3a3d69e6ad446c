#!/usr/bin/env python3
"""Development probe: force-kernel time of rank 0 when N bodies are range-partitioned over G ranks, for each
kernel variant, measured on ONE GPU (all G contexts on device 0, compute phases serialised with
NBODY_GROUP_SERIALIZE=1).  Used to pick the lanes-per-body factor K per own-range size.
    python scaling_probe.py [N] [variants]"""
import os
import sys

os.environ["NBODY_GROUP_SERIALIZE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 31, 50, 52, 54]
cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
bodies = nb.init_bodies(cfg)
print("N=%d radii 0; per-rank force kernel ms per step and the implied whole-job pairs/s" % n)
for world in (1, 2, 4, 8):
    for variant in variants:
        grp = nb.StepperGroup(world, cfg=cfg, kernel_variant=variant)
        grp.upload(bodies)
        grp.step(1)
        for r in grp.ranks:
            r.sync()
        s0 = [r.stats() for r in grp.ranks]
        for r in grp.ranks:
            r.set_kernel_timing(True)
        steps = 3
        grp.step(steps)
        s1 = [r.stats() for r in grp.ranks]
        ms = [(b.force_kernel_ms) / max(1, b.force_kernel_launches) for b in s1]
        pairs = sum(b.pairs - a.pairs for a, b in zip(s0, s1)) / steps
        print("G=%d variant=%2d  kernel ms/rank: max %.3f min %.3f  -> %.3e pairs/s" %
              (world, variant, max(ms), min(ms), pairs / (max(ms) * 1e-3)), flush=True)
        grp.close()
