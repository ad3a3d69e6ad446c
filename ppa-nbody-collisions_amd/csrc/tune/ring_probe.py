#!/usr/bin/env python3
"""Development probe for the ring kernel: force-kernel time per rank of a G-rank partition of N bodies on ONE GPU
(ranks serialised) for a list of kernel variants, plus the in-kernel phase stamps of the probe build (variant 58).
    python3 ring_probe.py N G variants [steps] [stock]"""
import os
import sys

os.environ["NBODY_GROUP_SERIALIZE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n, world = int(sys.argv[1]), int(sys.argv[2])
variants = [int(v) for v in sys.argv[3].split(",")]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kw = {} if (len(sys.argv) > 5 and sys.argv[5] == "stock") else {"minRadius": 0.0, "maxRadius": 0.0}
cfg = nb.stock_config(particleCount=n, **kw)
bodies = nb.init_bodies(cfg)
ref = None
for variant in variants:
    grp = nb.StepperGroup(world, cfg=cfg, kernel_variant=variant)
    grp.upload(bodies)
    grp.step(1)
    s0 = [r.stats() for r in grp.ranks]
    for r in grp.ranks:
        r.set_kernel_timing(True)
    grp.step(steps)
    s1 = [r.stats() for r in grp.ranks]
    ms = [b.force_kernel_ms / max(1, b.force_kernel_launches) for b in s1]
    pairs = sum(b.pairs - a.pairs for a, b in zip(s0, s1)) / steps
    out = grp.download()
    same = ""
    if ref is None:
        ref = out.block.copy()
    else:
        same = "  bits==first: %s" % np.array_equal(ref.view(np.uint32), out.block.view(np.uint32))
    print("N=%d G=%d variant=%2d  kernel ms/rank: max %.3f min %.3f  -> %.3e pairs/s%s" %
          (n, world, variant, max(ms), min(ms), pairs / (max(ms) * 1e-3), same), flush=True)
    if variant == 58:
        p = grp.ranks[0].ring_probe()
        turns = max(1, p[5])
        ghz = p[6] / max(1, p[7]) * 0.1
        print("   probe rank0: per wave-turn cycles: evaluate %.0f  wait %.0f  chain+publish %.0f  check %.0f;"
              " polls/turn %.2f; shader clock %.2f GHz (one wave's life %.3f ms)" %
              (p[0] / turns, p[1] / turns, p[2] / turns, p[3] / turns, p[4] / turns, ghz, p[7] / 1e5), flush=True)
    grp.close()
