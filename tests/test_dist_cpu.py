"""CPU test of the multi-rank PROTOCOL (gloo, world_size 2 and 3), no GPU.  This is an oracle-protocol test: it
validates the design of the sharded step, not the device code that implements it (compact_scatter / unpack_slots /
do_exchange run only on a GPU: tests/test_gpu_parity.py for the single-process group form, tests/test_gpu_multirank.py
and bench.py's parity leg for RCCL).  What IS product code here: the partition rule, nbody_partition.

The device code shards a step as: rank g owns a contiguous, block-aligned global range [lo_g, lo_g+cnt_g)
(nbody_partition); it computes the post-step state of its range (reading the full replica), compacts its survivors
into a slot {count, records, velocities}, the slots are all-gathered, and every rank rebuilds the replica
in rank order and RE-DRAWS the partition from the survivor count, taking the velocities of its new range from the
slots (csrc/nbody_kernels.hpp: compact_scatter, unpack_slots; csrc/nbody_partition.h: nbody_own_range_of; csrc/nbody_ctx.hip: nbody_step).
The slots are laid out for U = own_upper_of(n*) bodies, where n* is the body count FOUR STEPS BACK (the uploaded count
before that): the one bound every rank derives identically without looking at the current step (csrc/nbody_ctx.hip:
refresh_bound, slot_bodies); the all-gather moves that many rows, not the capacity.  This
file runs that protocol over torch.distributed with the CPU oracle doing the per-range arithmetic (the oracle is the
checker here, not a product path) and checks it against the single-rank oracle bit for bit - including deletions,
ranges that move between ranks, and the index-dependent literal semantics, which depend on GLOBAL indices and the
GLOBAL count.
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
    cap_own = ((n0 + 127) // 128 + world - 1) // world * 128
    lo, cnt = nb.partition(n0, rank, world)               # nbody_upload partition
    P, V, M, R = [a.copy() for a in ol.carve(full, n)]
    J = np.concatenate([P, M[:, None], R[:, None]], axis=1)          # replica {x,y,m,r}
    Vown = V[lo:lo + cnt].copy()
    moved = 0
    lag = 4                                                           # nbody_ctx::kLag
    history = []                                                      # body count after each step
    rows_moved = 0
    for s in range(steps):
        n_star = n0 if s < lag else history[s - lag]                  # refresh_bound: the count four steps back
        upper = ((n_star + 127) // 128 + world - 1) // world * 128    # slot_bodies = nbody_own_upper_of(n*, world)
        assert n <= n_star and cnt <= upper
        # compute phase on the own range, from the replica + own velocities
        blk = np.empty(6 * n, np.float32)
        p_, v_, m_, r_ = ol.carve(blk, n)
        p_[:] = J[:n, :2]; m_[:] = J[:n, 2]; r_[:] = J[:n, 3]
        v_[:] = 0
        v_[lo:lo + cnt] = Vown
        oP, oV, oM, oR, _, _ = ol.port_range(blk, n, lo, lo + cnt, dt, field, field, growth)
        keep = oM != 0                                                # src/nbody.cu:488-510
        slot = np.zeros((upper + 1, 6), np.float32)                   # row 0: header; then {x,y,m,r,vx,vy}
        c = int(keep.sum())
        assert c <= upper
        slot[0, 0] = c
        slot[1:1 + c] = np.concatenate([oP[keep], oM[keep, None], oR[keep, None], oV[keep]], axis=1)
        # exchange phase
        gathered = [torch.zeros(upper + 1, 6) for _ in range(world)]  # a rank with another `upper` would fail here
        dist.all_gather(gathered, torch.from_numpy(slot))
        rows_moved += world * (upper + 1)
        # commit phase: replica in rank order, partition re-drawn from the survivor count, own velocities from the slots
        counts = [int(g[0, 0]) for g in gathered]
        allrec = np.concatenate([g.numpy()[1:1 + k] for g, k in zip(gathered, counts)], axis=0)
        J = allrec[:, :4]
        n = sum(counts)
        old_lo = sum(counts[:rank])
        lo, cnt = nb.partition(n, rank, world)
        moved += int(lo != old_lo or cnt != counts[rank])
        assert cnt <= cap_own and lo % 128 == 0
        Vown = allrec[lo:lo + cnt, 4:6].copy()
        history.append(n)
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
        q.put((n, out.view(np.uint32).copy(), moved, rows_moved, steps * world * (cap_own + 1)))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_rule():
    """nbody_partition: block-aligned, contiguous, covering, level to within one reference block."""
    sys.path.insert(0, ROOT)
    import ppa_nbody_collisions_amd as nb
    for n in (0, 1, 2, 127, 128, 129, 255, 256, 1000, 1024, 65536, 65537, 262139, 262144, 1048576):
        for world in (1, 2, 3, 4, 7, 8):
            parts = [nb.partition(n, g, world) for g in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            assert all(parts[g][0] + parts[g][1] == parts[g + 1][0] for g in range(world - 1))
            assert all(lo % 128 == 0 for lo, c in parts if c > 0)
            full = [c for _, c in parts]
            assert max(full) - min(full) <= 128 + 127, (n, world, full)
            assert max(full) <= ((n + 127) // 128 + world - 1) // world * 128


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
    n_got, blk_got, moved, rows_moved, rows_at_capacity = q.get(timeout=240)
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
        assert n < n0          # the case really exercises deletions ...
        assert moved > 0       # ... and bodies changing hands when the partition is re-drawn
    if steps > 8 and field <= 5000:
        assert rows_moved < rows_at_capacity   # ... and the exchange shrinking with the count (four steps late)
