#!/usr/bin/env python3
"""Generates tests/golden/fma_pairs.npz: teacher-forced step pairs (S_t -> S_t+1) computed by the FMA-CONTRACTED
build of the literal shim (oracle/_ref/libnbody_ref_fma.so: the reference's own kernel text, src/nbody.cu:126-292,
compiled `g++ -O2 -ffp-contract=fast -mfma`).

Why: the reference's documented build is `nvcc -O3` (cudaCmd.txt:1), whose default -fmad=true contracts a*b+c into
FMAs - e.g. the collision predicate dx*dx + dy*dy (src/nbody.cu:131), the distance (include/vec2f.h:91-93), the force
accumulation (src/nbody.cu:239) and the drift (:288).  Which operations nvcc/ptxas would fuse is unknowable here (no
nvcc in this image), so the oracle of record is the NON-contracted evaluation and these vectors only BOUND the other
reading: tests assert that one step from the same S_t differs by <= 1e-5 (norm-wise, per array) and that the collision
outcomes (deleted set, absorbed masses) are identical.  Data only (inputs and expected outputs).

    make -C oracle ref && python tests/golden/make_golden_fma.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

DT, GROWTH, FIELD = np.float32(0.2), np.float32(0.1), 100000


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b.astype(np.float64)).max(), 1e-300))


def main():
    assert os.path.exists(ol.REF_FMA_SO) and ol.have_ref(), "make -C oracle ref"
    out = {"params": np.array([float(DT), float(GROWTH), FIELD, FIELD], dtype=np.float64)}
    # C1 (BASELINE configs[0]): the FMA build's OWN free-running trajectory at N=1024; pairs at five steps
    b = ol.ref_init(1024)
    n = 1024
    for t in range(100):
        keep = t in (0, 1, 9, 49, 99)
        before = b[:6 * n].copy()
        n_before = n
        n, pre = ol.ref_fma_step(b, n, DT, FIELD, FIELD, GROWTH, pre=keep)
        if keep:
            out["c1_t%d_in" % t] = before.view(np.uint32)
            out["c1_t%d_pre" % t] = pre.view(np.uint32).copy()
            out["c1_t%d_n" % t] = np.array([n_before, n], dtype=np.int32)
            # how far the non-contracted build of the same text is from it, from the same state
            nb_ = before.copy()
            n2, pre2 = ol.ref_step(nb_, n_before, DT, FIELD, FIELD, GROWTH, pre=True)
            P, V, M, R = ol.carve(pre, n_before)
            P2, V2, M2, R2 = ol.carve(pre2, n_before)
            print("c1 t=%d: n %d -> %d (no-FMA %d)  rel dP %.2e dV %.2e  masses equal %s" %
                  (t, n_before, n, n2, rel(P2, P), rel(V2, V), np.array_equal(M.view(np.uint32), M2.view(np.uint32))))
    print("c1 final n", n)
    # N=4096 stock, step 1
    b = ol.ref_init(4096)
    out["n4096_in"] = b.view(np.uint32).copy()
    n1, pre = ol.ref_fma_step(b, 4096, DT, FIELD, FIELD, GROWTH, pre=True)
    out["n4096_pre"] = pre.view(np.uint32).copy()
    out["n4096_n"] = np.array([4096, n1], dtype=np.int32)
    # N=65536 (C2 / C3 shapes), step 1: every 16th body, the deleted set, a hash of all masses
    for name, kw in (("n65536_stock", {}), ("n65536_r0", {"min_r": 0.0, "max_r": 0.0})):
        n0 = 65536
        b = ol.ref_init(n0, **kw)
        b0 = b.copy()
        n1, pre = ol.ref_fma_step(b, n0, DT, FIELD, FIELD, GROWTH, pre=True)
        P, V, M, R = ol.carve(pre, n0)
        idx = np.arange(0, n0, 16)
        out[name + "_kw"] = np.array([kw.get("min_r", 50.0), kw.get("max_r", 200.0)], dtype=np.float64)
        out[name + "_idx"] = idx.astype(np.int32)
        out[name + "_P"] = P[idx].view(np.uint32).copy()
        out[name + "_V"] = V[idx].view(np.uint32).copy()
        out[name + "_R"] = R[idx].view(np.uint32).copy()
        out[name + "_deleted"] = np.nonzero(M == 0)[0].astype(np.int32)
        out[name + "_mass_sha256"] = np.frombuffer(hashlib.sha256(M.tobytes()).digest(), dtype=np.uint8).copy()
        out[name + "_n"] = np.array([n0, n1], dtype=np.int32)
        out[name + "_maxabs"] = np.array([np.abs(P).max(), np.abs(V).max()], dtype=np.float64)
        n2, pre2 = ol.ref_step(b0, n0, DT, FIELD, FIELD, GROWTH, pre=True)
        P2, V2, M2, R2 = ol.carve(pre2, n0)
        print("%s: n %d -> %d (no-FMA %d)  rel dP %.2e dV %.2e  masses equal %s  radii rel %.2e" %
              (name, n0, n1, n2, rel(P2, P), rel(V2, V), np.array_equal(M.view(np.uint32), M2.view(np.uint32)),
               rel(R2, R)))
    np.savez_compressed(os.path.join(HERE, "fma_pairs.npz"), **out)


if __name__ == "__main__":
    main()
