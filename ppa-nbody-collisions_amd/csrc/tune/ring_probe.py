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
    if max(ms) > 1.1 * min(ms):
        print("   per rank:", " ".join("%.3f" % m for m in ms), flush=True)
    if variant == 58:
        p = grp.ranks[0].ring_probe()
        turns = max(1, p[5])
        ghz = p[6] / max(1, p[7]) * 0.1
        print("   probe rank0: per wave-turn cycles: evaluate %.0f  wait %.0f  chain+publish %.0f  check %.0f;"
              " polls/turn %.2f; shader clock %.2f GHz (one wave's life %.3f ms)" %
              (p[0] / turns, p[1] / turns, p[2] / turns, p[3] / turns, p[4] / turns, ghz, p[7] / 1e5), flush=True)
        ev = grp.ranks[0].events()
        if len(ev):
            t0 = ev["step"].astype(np.int64) & 0xffffffff
            t1 = ev["i"].astype(np.int64) & 0xffffffff
            base = t0.min()
            hw = ev["j"].astype(np.int64) & 0xffffffff
            xcc = (ev["kind"].astype(np.int64) >> 20) & 0xf
            cu = (hw >> 8) & 0xf
            se = (hw >> 13) & 0x7
            place = xcc * 1000 + se * 100 + cu
            print("   %d workgroup records (all launches); start ticks after first: p50 %d p90 %d max %d; "
                  "life ticks: min %d p50 %d max %d; distinct (xcc,se,cu): %d; max WGs on one CU: %d" %
                  (len(ev), np.percentile(t0 - base, 50), np.percentile(t0 - base, 90), (t0 - base).max(),
                   (t1 - t0).min(), np.percentile(t1 - t0, 50), (t1 - t0).max(), len(np.unique(place)),
                   np.bincount(np.unique(place, return_inverse=True)[1]).max()))
            last = ev[-min(len(ev), 512):]
            lt0 = (last["step"].astype(np.int64) & 0xffffffff); lt1 = (last["i"].astype(np.int64) & 0xffffffff)
            lb = lt0.min()
            print("   last launch: starts p50 %d p99 %d max %d; ends min %d p50 %d max %d (ticks after its first start)" %
                  (np.percentile(lt0 - lb, 50), np.percentile(lt0 - lb, 99), (lt0 - lb).max(), (lt1 - lb).min(),
                   np.percentile(lt1 - lb, 50), (lt1 - lb).max()))
            np.save("gpurun_out/r02_ring_wg_records_N%d_G%d.npy" % (n, world), ev)
    grp.close()
