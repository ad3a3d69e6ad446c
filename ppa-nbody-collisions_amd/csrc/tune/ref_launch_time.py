#!/usr/bin/env python3
"""Times the reference-shaped launches (nbody_launch_compute_forces_f32 + nbody_launch_move_bodies_f32, the drop-in
for the two <<<>>> sites src/nbody.cu:481-483) on a device block in the reference layout, per kernel behind them.
    python3 ref_launch_time.py [N] [reps] [stock]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ppa_nbody_collisions_amd as nb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kw = {} if "stock" in sys.argv[3:] else {"minRadius": 0.0, "maxRadius": 0.0}
cfg = nb.stock_config(particleCount=n, **kw)
bodies = nb.init_bodies(cfg)
host = torch.from_numpy(bodies.contiguousData.copy())
blocks = nb.lib.nbody_num_blocks(n)
for name, env in (("ring kernel through the launch workspace", {}), ("one-lane kernel on the block", {"NBODY_REF_LAUNCH_ONE_LANE": "1"})):
    os.environ["NBODY_REF_LAUNCH_ONE_LANE"] = env.get("NBODY_REF_LAUNCH_ONE_LANE", "0")
    dev = host.clone().cuda()
    upd_m = dev[4 * n:5 * n].clone()
    upd_r = dev[5 * n:6 * n].clone()
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        assert nb.lib.nbody_launch_compute_forces_f32(dev.data_ptr(), upd_m.data_ptr(), upd_r.data_ptr(), n, float(cfg.timestep),
                                                      cfg.fieldWidth, cfg.fieldHeight, blocks, float(cfg.growthRate), stream) == 0
        assert nb.lib.nbody_launch_move_bodies_f32(dev.data_ptr(), upd_m.data_ptr(), upd_r.data_ptr(), n, float(cfg.timestep),
                                                   blocks, stream) == 0
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    dev.copy_(host)                       # the timed launches start from the initial condition again (no compaction
    upd_m.copy_(dev[4 * n:5 * n])         # between them here: with collisions on, keep reps small)
    upd_r.copy_(dev[5 * n:6 * n])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    print("N=%d %s: %.3f ms per ComputeForces + MoveBodies" % (n, name, e0.elapsed_time(e1) / reps), flush=True)
nb.lib.nbody_launch_workspace_release()
