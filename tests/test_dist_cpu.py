"""CPU tests of the multi-rank protocol (gloo, world_size 2 and 3), no GPU.

The device code shards a step as: rank g owns the contiguous global range [lo_g, lo_g+cnt_g); it computes
the post-step state of its range (reading the full replica), compacts its survivors into a fixed-size slot
{count, records}, the slots are all-gathered, and every rank rebuilds the replica in rank order, taking
lo/cnt for the next step from the prefix of the counts (csrc/nbody_kernels.hpp: compact_scatter,
unpack_slots; csrc/nbody_ctx.hip: nbody_step).  This file runs exactly that protocol over torch.distributed
with the CPU oracle doing the per-range arithmetic (the oracle is the checker here, not a product path) and
checks it against the single-rank oracle bit for bit - including ragged ranges after deletions and the
index-dependent literal semantics, which depend on GLOBAL indices and the GLOBAL count.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, port, n0, field, steps, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle_lib as ol
    import ppa_nbody_collisions_amd as nb
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ol.port().oracle_set_threads(2)
    dt, growth = np.float32(0.2), np.float32(0.1)
    cfg = nb.stock_config(particleCount=n0, fieldWidth=field, fieldHeight=field)
    full = nb.init_bodies(cfg).contiguousData.copy()      # every rank uploads the full block
    n = n0
    cap_own = (n0 + world - 1) // world + 1
    lo = n0 * rank // world                               # nbody_upload partition
    cnt = n0 * (rank + 1) // world - lo
    P, V, M, R = [a.copy() for a in ol.carve(full, n)]
    J = np.concatenate([P, M[:, None], R[:, None]], axis=1)          # replica {x,y,m,r}
    Vown = V[lo:lo + cnt].copy()
    for s in range(steps):
        # compute phase on the own range, from the replica + own velocities
        blk = np.empty(6 * n, np.float32)
        p_, v_, m_, r_ = ol.carve(blk, n)
        p_[:] = J[:n, :2]; m_[:] = J[:n, 2]; r_[:] = J[:n, 3]
        v_[:] = 0
        v_[lo:lo + cnt] = Vown
        oP, oV, oM, oR, _, _ = ol.port_range(blk, n, lo, lo + cnt, dt, field, field, growth)
        keep = oM != 0                                                # src/nbody.cu:488-510
        slot = np.zeros((cap_own + 1, 4), np.float32)
        c = int(keep.sum())
        slot[0, 0] = c
        slot[1:1 + c] = np.concatenate([oP[keep], oM[keep, None], oR[keep, None]], axis=1)
        Vown = oV[keep]
        # exchange phase
        gathered = [torch.zeros(cap_own + 1, 4) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(slot))
        # commit phase
        counts = [int(g[0, 0]) for g in gathered]
        J = np.concatenate([g.numpy()[1:1 + k] for g, k in zip(gathered, counts)], axis=0)
        n = sum(counts)
        lo, cnt = sum(counts[:rank]), counts[rank]
    # assemble the full state on rank 0 (velocities by padded gather, like nbody_download)
    vbuf = np.zeros((cap_own, 2), np.float32)
    vbuf[:cnt] = Vown
    vg = [torch.zeros(cap_own, 2) for _ in range(world)]
    dist.all_gather(vg, torch.from_numpy(vbuf))
    cg = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(cg, torch.tensor([cnt]))
    if rank == 0:
        Vall = np.concatenate([v.numpy()[:int(k)] for v, k in zip(vg, cg)], axis=0)
        out = ol.make_block(J[:, :2], Vall, J[:, 2], J[:, 3])
        q.put((n, out.view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n0,field,steps", [(2, 1000, 5000, 12), (2, 1024, 5000, 8), (3, 700, 3000, 10),
                                                  (2, 300, 100000, 3)])
def test_sharded_protocol_equals_single_rank(world, n0, field, steps):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import ppa_nbody_collisions_amd as nb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:     # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n0, field, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    n_got, blk_got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = nb.stock_config(particleCount=n0, fieldWidth=field, fieldHeight=field)
    ref = nb.init_bodies(cfg).contiguousData.copy()
    n = n0
    for s in range(steps):
        n, *_ = ol.port_step(ref, n, np.float32(0.2), field, field, np.float32(0.1), want_events=False)
    assert n_got == n
    assert np.array_equal(blk_got, ref[:6 * n].view(np.uint32))
    if field <= 5000:
        assert n < n0          # the case really exercises deletions / ragged ranges
