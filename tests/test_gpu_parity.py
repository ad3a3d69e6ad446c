"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  * the committed golden vectors generated from the reference's own kernel text (tests/golden), and
  * the CPU oracle (oracle/nbody_oracle.c) on the same seeded inputs.
Bar: BIT-EXACT positions, velocities, masses, radii, survivor counts and collision event sets (fp32).  The
north_star tolerance is 1e-5 relative per step; the kernels keep the reference's accumulation order with
IEEE sqrt/divide and no contraction, so the tolerance used here is zero."""
import ctypes
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_FILES = sorted(glob.glob(os.path.join(GOLD, "steps_*.npz")))
DT, GROWTH = np.float32(0.2), np.float32(0.1)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def assert_bodies_equal(got, want_block, n, what=""):
    assert got.numBodies == n, (what, got.numBodies, n)
    g, w = bits(got.block), bits(np.asarray(want_block)[:6 * n])
    if not np.array_equal(g, w):
        bad = np.nonzero(g != w)[0]
        raise AssertionError("%s: %d of %d words differ, first at %s" % (what, len(bad), len(g), bad[:8]))


def test_ieee_sqrt_and_reciprocal_exhaustive(nb):
    """All 2^32 fp32 inputs: the kernels' sqrt and 1/x are correctly rounded (what the oracle's x86 sqrtss /
    divss compute)."""
    m = (ctypes.c_uint64 * 3)()
    assert nb.lib.nbody_selftest_ieee_f32(0, ctypes.byref(m)) == 0, nb.lib.nbody_last_error_string()
    assert (m[0], m[1], m[2]) == (0, 0, 0)


def test_lds_record_is_never_torn(nb):
    """The ring kernel's hand-off assumption: a lane's 16-byte LDS record (ds_write_b128 / ds_read_b128) is seen
    entirely old or entirely new by another wave.  512 workgroups x 64 records x 20000 rewrites, 7 polling waves."""
    r = (ctypes.c_uint64 * 3)()
    assert nb.lib.nbody_selftest_lds_record(0, 20000, ctypes.byref(r)) == 0, nb.lib.nbody_last_error_string()
    assert r[0] == 0 and r[1] == 0, (r[0], r[1])
    assert r[2] >= 512 * 7 * 64


def test_ring_handoff_timeout_is_loud(nb, monkeypatch):
    """A ring hand-off wait that gives up must not pass as a result (CUDA_SYNC_CHECK convention,
    src/nbody.cu:20-33): with the poll limit shrunk to one poll some wave of the ring gives up, the kernel poisons
    the chain, and every host entry point that looks at the device returns NBODY_ERR_HIP until the next upload."""
    monkeypatch.setenv("NBODY_RING_SPIN_LIMIT", "1")
    cfg = nb.stock_config(particleCount=16384, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg, kernel_variant=50)
    monkeypatch.delenv("NBODY_RING_SPIN_LIMIT")
    st.upload(bodies)
    st.step(2)
    for call in (st.sync, st.download, st.stats, st.body_count, lambda: st.step(1),
                 lambda: st.save_state("/tmp/never_written.nbody"), lambda: st.render_image(64, 64)):
        with pytest.raises(nb.NbodyError) as e:
            call()
        assert e.value.status == -6 and "hand-off" in str(e.value), str(e.value)
    assert not os.path.exists("/tmp/never_written.nbody")
    # the poisoned state is visibly not a result; a fresh upload clears the failure but the limit stays
    st.close()
    ok = nb.Stepper(cfg, kernel_variant=50)              # default limit again
    ok.upload(bodies)
    ok.step(2)
    one = nb.Stepper(cfg, kernel_variant=31)
    one.upload(bodies)
    one.step(2)
    a, b = ok.download(), one.download()
    assert a.numBodies == b.numBodies and np.array_equal(bits(a.block), bits(b.block))
    ok.close(); one.close()


@pytest.mark.parametrize("what,log", [("t", True), ("q", True), ("q", False)])
def test_failed_index_check_is_reported_not_trusted(nb, monkeypatch, what, log):
    """The ring kernel checks the indices it forms itself (the staging index of the epilogue in every build; in the
    event-logging builds also the source range of every window gather and the radius-bound lookups).  Told - by a testing
    aid - that the tiled copy is one tile long (`t`) or that the staging arrays hold one body (`q`), the checks fail: the
    accesses are skipped, nothing faults, and every host call that looks at the device returns NBODY_ERR_HIP naming
    the failed checks (CUDA_SYNC_CHECK's role, src/nbody.cu:20-33) until the next upload."""
    cfg = nb.stock_config(particleCount=4096, fieldWidth=20000, fieldHeight=20000)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg, record_events=log)
    st.upload(bodies)
    monkeypatch.setenv("NBODY_TEST_INDEX_CHECKS", what)
    st.step(1)
    monkeypatch.delenv("NBODY_TEST_INDEX_CHECKS")
    for call in (st.sync, st.download, lambda: st.step(1)):
        with pytest.raises(nb.NbodyError) as e:
            call()
        assert e.value.status == -6 and "index check" in str(e.value) and " 0 failed index" not in str(e.value), str(e.value)
    st.upload(bodies)                                  # a fresh upload clears it
    st.step(2)
    one = nb.Stepper(cfg, kernel_variant=31)
    one.upload(bodies)
    one.step(2)
    a, b = st.download(), one.download()
    assert a.numBodies == b.numBodies and np.array_equal(bits(a.block), bits(b.block))
    st.close(); one.close()


def test_fp64_fast_chain_against_ieee(nb):
    """2^32 inputs of each of three families of the guarded domain: the fp64 force kernel's sqrt / 1/d^3 chain
    gives the bits of the compiler's correctly-rounded sqrt and divide (not exhaustive: fp64 cannot be)."""
    m = (ctypes.c_uint64 * 2)()
    assert nb.lib.nbody_selftest_chain_f64(0, 1 << 32, ctypes.byref(m)) == 0, nb.lib.nbody_last_error_string()
    assert (m[0], m[1]) == (0, 0)


def test_fp64_reciprocal_of_all_ones_significand(nb):
    """The one known exception of Newton-type fp64 reciprocals (Markstein): c = (2 - 2^-52) 2^k.  The last fma of the
    refinement sees an exact tie and rounds one ulp low.  On every exponent of the guarded domain: the general code's
    reciprocal (ieee_rcp: closed form for that significand) must be right, and the fp64 kernel's screen (low word of c
    == 0xffffffff -> the chunk is redone by the general code) must catch every such c.  Informational: whether the
    compiler's bare division and the fast refinement get them right (they need not: that is why the screen exists)."""
    r = (ctypes.c_uint64 * 5)()
    assert nb.lib.nbody_selftest_rcp_ones_f64(0, ctypes.byref(r)) == 0, nb.lib.nbody_last_error_string()
    print("\nall-ones significands: %d inputs; general code wrong on %d, bare 1.0/c wrong on %d, fast refinement wrong on "
          "%d, missed by the screen %d" % (r[4], r[0], r[1], r[2], r[3]))
    assert r[4] == 1501 and r[0] == 0 and r[3] == 0


def _stepper(nb, cap, fw, fh, dt=DT, growth=GROWTH, **kw):
    return nb.Stepper(capacity=cap, timestep=float(dt), growthRate=float(growth), fieldWidth=fw, fieldHeight=fh,
                      **kw)


# automatic | v1 | v3 K=1,2,4,8 | v3 256-thread | ring: 2x8, 4x4, 1x8 rings x waves per workgroup, 16-position turns
VARIANTS = [0, 1, 11, 12, 14, 18, 31, 32, 50, 52, 53, 54]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", STEP_FILES, ids=[os.path.basename(p)[6:-4] for p in STEP_FILES])
def test_golden_free_run(nb, path, variant):
    z = np.load(path)
    dt, growth, fw, fh = z["params"]
    n0 = int(z["n0"])
    counts = z["counts"]
    st = _stepper(nb, n0, int(fw), int(fh), np.float32(dt), np.float32(growth), kernel_variant=variant)
    st.upload(nb.BodiesData.from_block(z["init"].view(np.float32), n0))
    for s in range(1, len(counts) + 1):
        st.step(1)
        if "after_%d" % s in z:
            assert_bodies_equal(st.download(), z["after_%d" % s].view(np.float32), int(counts[s - 1]),
                                "%s step %d" % (os.path.basename(path), s))
        else:
            assert st.body_count() == counts[s - 1]
    # pair counter = what the oracle says the stepper evaluates
    ns = [n0] + [int(c) for c in counts[:-1]]
    assert st.stats().pairs == sum(ol.port().oracle_pairs_per_step(n, ol.LITERAL) for n in ns)
    st.close()


def _rel(a, b, scale=None):
    scale = float(np.abs(np.asarray(b, np.float64)).max()) if scale is None else scale
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(scale, 1e-300))


def test_bounded_against_fma_contracted_build(nb):
    """The other plausible compile of the reference (`nvcc -O3`, cudaCmd.txt:1: -fmad=true would contract
    src/nbody.cu:131,232,239,288 and include/vec2f.h:91-93 into FMAs).  tests/golden/fma_pairs.npz holds teacher-forced
    pairs S_t -> S_t+1 from an FMA-contracted build of the reference's kernel text.  The HIP path (no contraction,
    bit-identical to the oracle of record) loads S_t, steps ONCE and must be within the north_star tolerance of that
    build - max|dP|/max|P| and max|dV|/max|V| <= 1e-5 - with IDENTICAL collision outcomes: survivor count, deleted set
    D_t, and bit-identical masses (same absorb set E_t: a body's new mass is the ordered sum of what it absorbed).
    Measured margin: 2e-7 at worst, printed below."""
    z = np.load(os.path.join(GOLD, "fma_pairs.npz"))
    dt, growth, fw, fh = z["params"]
    worst = {}
    for key in [k[:-3] for k in z.files if k.endswith("_in")]:
        n0, n1 = (int(v) for v in z[key + "_n"])
        st = _stepper(nb, n0, int(fw), int(fh), np.float32(dt), np.float32(growth), record_events=True)
        st.upload(nb.BodiesData.from_block(z[key + "_in"].view(np.float32), n0))
        st.step(1)
        out = st.download()
        ev = st.events()
        st.close()
        fP, fV, fM, fR = ol.carve(z[key + "_pre"].view(np.float32), n0)
        keep = fM != 0
        assert out.numBodies == n1 == int(keep.sum()), key
        assert np.array_equal(np.unique(ev["i"][ev["kind"] == 1]), np.nonzero(~keep)[0]), key        # D_t
        assert np.array_equal(bits(out.Masses), bits(fM[keep])), key                                   # same E_t
        worst[key] = (_rel(out.Positions, fP[keep]), _rel(out.Velocities, fV[keep]), _rel(out.Radii, fR[keep]))
        assert worst[key][0] <= 1e-5 and worst[key][1] <= 1e-5 and worst[key][2] <= 1e-6, (key, worst[key])
    for key in ("n65536_stock", "n65536_r0"):                          # C3 / C2 shapes, step 1
        n0, n1 = (int(v) for v in z[key + "_n"])
        min_r, max_r = (float(v) for v in z[key + "_kw"])
        cfg = nb.stock_config(particleCount=n0, minRadius=min_r, maxRadius=max_r)
        st = nb.Stepper(cfg, record_events=True, event_capacity=1 << 22)
        st.upload(nb.init_bodies(cfg))
        st.step(1)
        out = st.download()
        ev = st.events(cap=1 << 22)
        st.close()
        deleted = np.unique(ev["i"][ev["kind"] == 1]).astype(np.int64)
        assert out.numBodies == n1, key
        assert np.array_equal(deleted, z[key + "_deleted"]), key                                       # D_t
        pre_m = np.zeros(n0, np.float32)                               # pre-compaction masses: 0 where deleted
        alive = np.ones(n0, bool)
        alive[deleted] = False
        pre_m[alive] = out.Masses
        assert hashlib.sha256(pre_m.tobytes()).digest() == z[key + "_mass_sha256"].tobytes(), key      # same E_t
        idx = z[key + "_idx"].astype(np.int64)
        idx = idx[alive[idx]]
        sel = np.isin(z[key + "_idx"], idx)
        post = idx - np.searchsorted(deleted, idx)
        maxP, maxV = z[key + "_maxabs"]
        worst[key] = (_rel(out.Positions[post], z[key + "_P"].view(np.float32)[sel], maxP),
                      _rel(out.Velocities[post], z[key + "_V"].view(np.float32)[sel], maxV),
                      _rel(out.Radii[post], z[key + "_R"].view(np.float32)[sel], 200.0))
        assert worst[key][0] <= 1e-5 and worst[key][1] <= 1e-5 and worst[key][2] <= 1e-6, (key, worst[key])
    print("\nHIP path vs FMA-contracted build, one step, (rel dP, rel dV, rel dR):",
          {k: tuple("%.1e" % v for v in w) for k, w in worst.items()})
    assert max(max(w[:2]) for w in worst.values()) < 1e-6


def test_golden_multi_step_enqueue(nb):
    """100 steps enqueued in one call (no host synchronisation in between) = C1 of BASELINE.json."""
    z = np.load(os.path.join(GOLD, "steps_c1_n1024.npz"))
    st = _stepper(nb, 1024, 100000, 100000)
    st.upload(nb.BodiesData.from_block(z["init"].view(np.float32), 1024))
    st.step(100)
    assert_bodies_equal(st.download(), z["after_100"].view(np.float32), 977, "c1 100 steps")
    st.close()


@pytest.mark.parametrize("n,field,steps", [(1000, 5000, 8), (1024, 5000, 8), (300, 2000, 6), (4096, 20000, 3)])
def test_events_match_oracle(nb, n, field, steps):
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg, record_events=True)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(steps):
        st.step(1)
        cur2, stats, ab, de, _ = ol.port_step(blk, cur, DT, field, field, GROWTH)
        ev = st.events()
        ev = ev[ev["step"] == s]
        got_abs = sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0])
        assert got_abs == sorted((int(a), int(b)) for a, b in ab), "E_t step %d" % s
        assert sorted(set(int(e["i"]) for e in ev[ev["kind"] == 1])) == sorted(int(d) for d in de), "D_t %d" % s
        cur = cur2
        assert_bodies_equal(st.download(), blk, cur, "step %d" % s)
    st.close()


def test_random_small_cases_match_oracle(nb):
    """Seeded random sweep over the awkward part of the parameter space: N in 1..700 (all the N < 128,
    129..255 and non-multiple-of-128 quirks, counts that shrink across those boundaries), dense fields, random
    radius / mass ranges, literal and clean semantics, fp32 and fp64, every kernel variant, 1..4 ranks.  Bit-exact against the
    oracle after every step."""
    rng = np.random.default_rng(20240611)
    for case in range(150):
        n = int(rng.choice([rng.integers(1, 128), rng.integers(128, 260), rng.integers(260, 700)]))
        field = int(rng.choice([300, 1000, 3000, 20000]))
        min_r = float(rng.choice([0.0, 1.0, 10.0]))
        max_r = min_r + float(rng.choice([0.0, 5.0, 60.0]))
        max_m = float(rng.choice([1e5, 1e12, 1e17]))
        sem = int(rng.integers(0, 2))
        variant = int(rng.choice(VARIANTS))
        world = int(rng.integers(1, 5))
        f64 = bool(rng.integers(0, 5) == 0)
        cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field, minRadius=min_r,
                              maxRadius=max_r, maxRandBodyMass=max_m)
        bodies = nb.init_bodies(cfg, nb.F64 if f64 else nb.F32)
        bodies.Velocities[:] = rng.uniform(-50, 50, size=(n, 2)).astype(bodies.Velocities.dtype)   # walls matter
        grp = nb.StepperGroup(world, cfg=cfg, semantics=sem, kernel_variant=variant,
                              precision=nb.F64 if f64 else nb.F32)
        grp.upload(bodies)
        blk = bodies.contiguousData.copy()
        cur = n
        dt, gr = (float(DT), float(GROWTH)) if f64 else (DT, GROWTH)
        what = "case %d: n=%d field=%d r=[%g,%g] m<=%g sem=%d variant=%d world=%d f64=%d" % (
            case, n, field, min_r, max_r, max_m, sem, variant, world, f64)
        for s in range(5):
            grp.step(1)
            cur, *_ = ol.port_step(blk, cur, dt, field, field, gr, semantics=sem, want_events=False)
            assert_bodies_equal(grp.download(), blk, cur, "%s step %d" % (what, s))
            if cur == 0:
                break
        grp.close()


def test_random_medium_cases_match_oracle(nb):
    """Seeded random sweep at sizes where most tiles take the fast evaluation path (N 2 000..20 000, not multiples
    of 128, collisions scattered through the walk so flagged chunks / sub-tiles are redone by the general code)."""
    rng = np.random.default_rng(77)
    for case in range(14):
        n = int(rng.integers(2000, 20000))
        field = int(rng.choice([20000, 60000, 100000]))
        max_r = float(rng.choice([0.0, 30.0, 200.0]))
        sem = int(rng.integers(0, 2))
        variant = int(rng.choice(VARIANTS))
        world = int(rng.choice([1, 1, 2, 3]))
        cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field, minRadius=0.0, maxRadius=max_r)
        bodies = nb.init_bodies(cfg)
        grp = nb.StepperGroup(world, cfg=cfg, semantics=sem, kernel_variant=variant)
        grp.upload(bodies)
        blk = bodies.contiguousData.copy()
        cur = n
        for s in range(3):
            grp.step(1)
            cur, *_ = ol.port_step(blk, cur, DT, field, field, GROWTH, semantics=sem, want_events=False)
            assert_bodies_equal(grp.download(), blk, cur, "case %d: n=%d field=%d max_r=%g sem=%d variant=%d "
                                "world=%d step %d" % (case, n, field, max_r, sem, variant, world, s))
        grp.close()


def _nan_aware_equal(got, want):
    g, w = np.asarray(got), np.asarray(want)
    both_nan = np.isnan(g) & np.isnan(w)
    return np.array_equal(bits(g)[~both_nan.ravel()], bits(w)[~both_nan.ravel()])


@pytest.mark.parametrize("variant", VARIANTS)
def test_extreme_values_take_the_general_path(nb, variant):
    """Non-finite / huge / coincident / denormally close bodies: the fast evaluation chain must hand these to
    the general code (per-tile coordinate bound, per-wave own-coordinate bound, per-pair distance flag).
    NaN payloads differ between x86 and gfx950, so NaNs compare equal to NaNs."""
    n = 2048
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    P, V, M, R = bodies.Positions, bodies.Velocities, bodies.Masses, bodies.Radii
    P[100] = [1e20, -3e25]            # beyond the 2^38 coordinate bound
    P[300] = [np.inf, 5.0]
    P[301] = [np.nan, 7.0]
    P[500] = P[499]                   # coincident, radii 0: d2 == 0 <= 0 -> collision
    P[700] = P[699] + np.float32([1e-30, 0])        # rounds to coincident
    P[900] = [3e-25, 1e-26]
    P[901] = [3e-25 + 1e-31, 1e-26]   # denormally small separation
    P[1100] = [1.0e-3, 0]
    P[1101] = [1.0e-3 + 2.0e-11, 0]
    M[1300] = np.nan
    M[1301] = np.inf
    R[1500] = np.inf
    R[1501] = np.nan
    V[1700] = [1e30, -1e30]
    st = nb.Stepper(cfg, kernel_variant=variant)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(4):
        st.step(1)
        cur, *_ = ol.port_step(blk, cur, DT, 100000, 100000, GROWTH, want_events=False)
        out = st.download()
        assert out.numBodies == cur, "step %d" % s
        assert _nan_aware_equal(out.block, blk[:6 * cur]), "step %d" % s
    st.close()


@pytest.mark.parametrize("variant,precision", [(0, 0), (50, 0), (52, 0), (53, 0), (54, 0), (11, 0), (14, 0), (31, 0),
                                               (0, 1), (1, 1)])
@pytest.mark.parametrize("small", [False, True], ids=["nan-screen", "per-pair-screen"])
def test_coincident_bodies_at_zero_radii(nb, variant, precision, small):
    """All radii +0 and every coordinate in [2^-16, 2^38): the ring kernel runs WITHOUT a per-pair screen - the only pairs
    it must not add up are coincident bodies (d2 == 0 <= 0: a collision, src/nbody.cu:215-226), and those make their term
    NaN, which the wave sees in the sum after the turn's adds (kCoordFloor).  Coincident pairs inside a tile, across tiles,
    across the wrap, three bodies on one point, equal masses (both absorb), the self position next to a twin; events
    (E_t, D_t) and the whole state against the oracle over several steps.  `small`: one coordinate below 2^-16 switches
    the launch back to the v_min3 screen (Meta::summary bit 2) - same results.  The one-lane kernels (fp64's production
    kernel among them) do the same per 32-position chunk of a tile (FastDomain::floor)."""
    n = 4096
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg, precision)
    P, M = bodies.Positions, bodies.Masses
    dt, gr = (float(DT), float(GROWTH)) if precision == nb.F64 else (DT, GROWTH)
    u = np.uint64 if precision == nb.F64 else np.uint32
    P[70] = P[5]                      # same tile, same wave
    P[200] = P[130]                   # same tile, other half
    P[1000] = P[300]                  # across tiles
    P[4090] = P[3]                    # across the wrap (last tile / tile 0)
    P[2000] = P[2001] = P[2002]       # three on one point
    P[2500] = P[2600]
    M[2500] = M[2600]                 # equal masses: both absorb, neither is deleted
    P[129] = P[128]                   # the twin sits at walk position 1 of its partner, right after the self position
    if small:
        P[3500, 1] = 1e-6 if precision == nb.F32 else 1e-60
    st = nb.Stepper(cfg, kernel_variant=variant, record_events=True, precision=precision)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(4):
        st.clear_events()
        st.step(1)
        n_before = cur
        cur, _, ab, de, _ = ol.port_step(blk, cur, dt, 100000, 100000, gr)
        out = st.download()
        assert out.numBodies == cur, "step %d" % s
        assert np.array_equal(out.block.view(u), blk[:6 * cur].view(u)), "step %d" % s
        ev = st.events()
        assert sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0]) == sorted((int(a), int(b)) for a, b in ab), s
        assert sorted(set(int(e["i"]) for e in ev[ev["kind"] == 1])) == sorted(int(d) for d in de), s
        if s == 0:
            assert len(ab) >= 8 and n_before - cur >= 6, (len(ab), n_before - cur)
    st.close()


@pytest.mark.parametrize("variant", [0, 50, 52, 54, 31])
@pytest.mark.parametrize("masses", ["finite", "non-finite"])
def test_non_finite_sums_under_the_nan_screen(nb, variant, masses):
    """The NaN-sum screen (all radii +0, all coordinates in [2^-16, 2^38), all masses below 2^90) flags a lane whenever its
    running sum is NaN.  `finite`: three heavy bodies (2^89) one float spacing apart make terms overflow - +inf from one
    side, -inf from the other, NaN sums for them and inf for their neighbours in the walk - with every mass finite, so the
    screen stays on and those lanes take the exact path in every later turn (slow, and it must still be right); a coincident
    pair sits among them.  `non-finite`: a NaN and two infinite masses switch the launch back to the per-pair screen
    (Meta::summary bit 3): same oracle.  NaN payloads differ between x86 and gfx950: NaNs compare equal to NaNs."""
    n = 2048
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    P, M = bodies.Positions, bodies.Masses
    if masses == "finite":
        one = np.float32(1.0)
        P[600] = [np.nextafter(one, np.float32(0)), 3.0]
        P[601] = [one, 3.0]
        P[1300] = [np.nextafter(one, np.float32(2)), 3.0]
        M[600] = M[601] = M[1300] = np.float32(2.0 ** 89)
    else:
        M[700] = np.nan
        M[1500] = np.inf
        M[1501] = -np.inf
    P[900] = P[40]
    st = nb.Stepper(cfg, kernel_variant=variant)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(3):
        st.step(1)
        cur, *_ = ol.port_step(blk, cur, DT, 100000, 100000, GROWTH, want_events=False)
        out = st.download()
        assert out.numBodies == cur, "step %d" % s
        assert _nan_aware_equal(out.block, blk[:6 * cur]), "step %d" % s
    if masses == "finite":
        assert not np.isfinite(blk[:6 * cur]).all(), "the case is meant to produce non-finite state"
    st.close()


@pytest.mark.parametrize("variant", [0, 50, 52, 54, 31])
@pytest.mark.parametrize("n", [3000, 4096])
def test_collision_screen_radius_bounds(nb, variant, n):
    """The ring kernel screens a turn for collisions with ONE threshold per lane, fma(R, R, 2^-80), R = |ri| + the largest
    |radius| of the aligned 128-body tiles the window touches (kept per tile by nbody_upload / unpack_slots); flagged
    lanes then get the exact status of their pairs.  Bounded coordinates, so that path is the one in use.  A giant, a
    negative, a NaN, an infinite and a denormal radius, a giant in the last (partial, wrapped-into) tile, a NaN mass
    inside a giant's reach (a hit the reference's if / else-if does not handle: the force term stays), over enough steps
    for deletions to move bodies from tile to tile; N = 3000 makes windows straddle two tiles and wrap."""
    field = 20000
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    P, M, R = bodies.Positions, bodies.Masses, bodies.Radii
    R[10] = 3000.0
    P[12] = P[10] + np.float32([100.0, 0.0])
    M[12] = np.nan
    R[200] = -400.0
    R[777] = np.nan
    R[1500] = 1e-42
    R[2100] = np.inf
    R[n - 1] = 2500.0
    R[130:140] = 60.0
    st = nb.Stepper(cfg, kernel_variant=variant, record_events=True, event_capacity=1 << 20)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(5):
        st.step(1)
        cur, stats, ab, de, _ = ol.port_step(blk, cur, DT, field, field, GROWTH)
        ev = st.events()
        ev = ev[ev["step"] == s]
        assert sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0]) == \
            sorted((int(x), int(y)) for x, y in ab), "E_t step %d" % s
        assert sorted(set(int(e["i"]) for e in ev[ev["kind"] == 1])) == sorted(int(x) for x in de), "D_t step %d" % s
        out = st.download()
        assert out.numBodies == cur and _nan_aware_equal(out.block, blk[:6 * cur]), "step %d" % s
    st.close()


@pytest.mark.parametrize("variant", [0, 11, 50, 52])
def test_unbounded_tile_mid_walk(nb, variant):
    """A tile with an out-of-range coordinate in the MIDDLE of every other body's walk (not in their own block):
    the fast path must hand exactly that tile to the general code and resume, for all kernels (this is the
    non-fast-turn-between-fast-turns path of the ring kernel: with an unbounded body in the replica it scans
    every window, Meta::summary)."""
    n = 4096
    cfg = nb.stock_config(particleCount=n, fieldWidth=30000, fieldHeight=30000)
    bodies = nb.init_bodies(cfg)
    bodies.Positions[1500] = [3e12, -7e11]
    bodies.Positions[2700] = [np.inf, 1.0]
    st = nb.Stepper(cfg, kernel_variant=variant, record_events=True)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(3):
        st.step(1)
        cur, stats, ab, de, _ = ol.port_step(blk, cur, DT, 30000, 30000, GROWTH)
        ev = st.events()
        ev = ev[ev["step"] == s]
        assert sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0]) == \
            sorted((int(x), int(y)) for x, y in ab), "E_t step %d" % s
        out = st.download()
        assert out.numBodies == cur and _nan_aware_equal(out.block, blk[:6 * cur]), "step %d" % s
    st.close()


def test_big_golden_n65536(nb):
    """One literal step at N=65536 (configs[1], configs[2] shapes) against the sha256 of the reference's own
    kernel text's output."""
    g = json.load(open(os.path.join(GOLD, "big_n65536.json")))
    for name, kw in (("stock_radii", {}), ("radii0", {"minRadius": 0.0, "maxRadius": 0.0})):
        cfg = nb.stock_config(particleCount=65536, **kw)
        st = nb.Stepper(cfg)
        st.upload(nb.init_bodies(cfg))
        st.step(1)
        out = st.download()
        assert out.numBodies == g[name]["n1"]
        assert hashlib.sha256(out.block.tobytes()).hexdigest() == g[name]["sha256_post"], name
        st.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_group_equals_single(nb, world):
    """Range partition over `world` contexts on one GPU, per-step slot exchange, ragged ranges after
    deletions: bit-identical to the single-context run and to the golden."""
    z = np.load(os.path.join(GOLD, "steps_dense_n1000.npz"))
    dt, growth, fw, fh = z["params"]
    grp = nb.StepperGroup(world, capacity=1000, timestep=float(dt), growthRate=float(growth), fieldWidth=int(fw),
                          fieldHeight=int(fh))
    grp.upload(nb.BodiesData.from_block(z["init"].view(np.float32), 1000))
    done = 0
    for s in (1, 5, 40):
        grp.step(s - done)
        done = s
        assert_bodies_equal(grp.download(), z["after_%d" % s].view(np.float32), int(z["counts"][s - 1]),
                            "world %d step %d" % (world, s))
    los = [r.own_range() for r in grp.ranks]
    assert los[0][0] == 0 and all(los[k][0] + los[k][1] == los[k + 1][0] for k in range(world - 1))
    grp.close()


@pytest.mark.parametrize("case", ["three_body", "edge_n1", "edge_n2", "edge_n129", "edge_n130"])
def test_more_ranks_than_work(nb, case):
    """8 ranks over 1, 2, 3 bodies (most ranks own nothing) and over the N=129/130 quirk cases."""
    z = np.load(os.path.join(GOLD, "steps_%s.npz" % case))
    dt, growth, fw, fh = z["params"]
    n0 = int(z["n0"])
    grp = nb.StepperGroup(8, capacity=n0, timestep=float(dt), growthRate=float(growth), fieldWidth=int(fw),
                          fieldHeight=int(fh))
    grp.upload(nb.BodiesData.from_block(z["init"].view(np.float32), n0))
    for s in range(1, len(z["counts"]) + 1):
        grp.step(1)
        if "after_%d" % s in z:
            assert_bodies_equal(grp.download(), z["after_%d" % s].view(np.float32), int(z["counts"][s - 1]),
                                "%s step %d" % (case, s))
    assert sum(r.own_range()[1] for r in grp.ranks) == int(z["counts"][-1])
    grp.close()


@pytest.mark.parametrize("world", [2, 8])
def test_big_golden_n65536_sharded(nb, world):
    """The N=65536 stock-radii golden (10 095 deletions in one step) through a `world`-rank partition with the
    AUTOMATIC kernel choice: 8 ranks own 8192 bodies each, which selects the ring kernel with one ring per workgroup
    (2 ranks: 32768 each, two rings per workgroup)."""
    g = json.load(open(os.path.join(GOLD, "big_n65536.json")))["stock_radii"]
    cfg = nb.stock_config(particleCount=65536)
    grp = nb.StepperGroup(world, cfg=cfg)
    grp.upload(nb.init_bodies(cfg))
    grp.step(1)
    out = grp.download()
    assert out.numBodies == g["n1"]
    assert hashlib.sha256(out.block.tobytes()).hexdigest() == g["sha256_post"]
    grp.step(2)                      # ragged ranges now; compare with the single-rank run
    one = nb.Stepper(cfg)
    one.upload(nb.init_bodies(cfg))
    one.step(3)
    a, b = grp.download(), one.download()
    assert a.numBodies == b.numBodies and np.array_equal(bits(a.block), bits(b.block))
    grp.close(); one.close()


def test_headline_size_eight_ranks_equals_one(nb):
    """BASELINE.json metric shape: N=262144 range-partitioned over 8 ranks (automatic kernel choice = the ring
    kernel with 2 rings x 8 waves per workgroup on 32768 own bodies; 1 rank: 4 rings x 4 waves) against the
    single-rank run, whole state, bit for bit."""
    cfg = nb.stock_config(particleCount=262144, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    grp = nb.StepperGroup(8, cfg=cfg)
    grp.upload(bodies)
    grp.step(2)
    one = nb.Stepper(cfg)
    one.upload(bodies)
    one.step(2)
    a, b = grp.download(), one.download()
    assert a.numBodies == b.numBodies
    assert np.array_equal(bits(a.block), bits(b.block))
    assert sum(r.stats().pairs for r in grp.ranks) == one.stats().pairs
    grp.close(); one.close()


@pytest.mark.parametrize("radii", ["radii0", "stock"])
def test_c2_c3_thousand_steps_all_paths_agree(nb, radii):
    """BASELINE.json configs[1]/[2]: N=65536, 1000 steps, without and with collisions.  The oracle cannot run
    this horizon in test time; what is checked is (1) step 1 against the golden sha256 from the reference,
    and (2) after all 1000 steps the automatic kernel (ring), the one-lane-per-body kernel, the first-generation
    kernel and a 4-rank range partition hold the same state bit for bit (three force kernels that share only the
    pair function), with the survivor count shrinking through the kernel-selection and ragged-N regimes.
    NOTE: beyond step 1 this is a CROSS-VARIANT check, not an oracle comparison - the oracle-anchored horizons are
    100 steps at N=1024 and 1000 steps at N=2048 (tests/golden/steps_c1_n1024, steps_long_n2048)."""
    g = json.load(open(os.path.join(GOLD, "big_n65536.json")))["radii0" if radii == "radii0" else "stock_radii"]
    kw = {"minRadius": 0.0, "maxRadius": 0.0} if radii == "radii0" else {}
    cfg = nb.stock_config(particleCount=65536, **kw)
    bodies = nb.init_bodies(cfg)
    states = {}
    for name, variant in (("auto", 0), ("one-lane", 31), ("v1", 1)):
        st = nb.Stepper(cfg, kernel_variant=variant)
        st.upload(bodies)
        st.step(1)
        out = st.download()
        assert out.numBodies == g["n1"]
        assert hashlib.sha256(out.block.tobytes()).hexdigest() == g["sha256_post"], name
        st.step(999)
        states[name] = st.download()
        st.close()
    grp = nb.StepperGroup(4, cfg=cfg)
    grp.upload(bodies)
    grp.step(1000)
    states["group4"] = grp.download()
    grp.close()
    ref = states["auto"]
    if radii == "stock":
        assert ref.numBodies < g["n1"]
    for name, out in states.items():
        assert out.numBodies == ref.numBodies, name
        assert np.array_equal(bits(out.block), bits(ref.block)), name


def test_headline_thousand_steps_partitions_agree(nb):
    """BASELINE.json configs[3] (C4) over its own horizon: N=262144, radii 0, 1000 steps; the single-rank run (ring
    kernel, 4 rings x 4 waves per workgroup) and the 8-rank partition (2 rings x 8 waves on 32768 own bodies, slots
    exchanged and the partition re-drawn 1000 times) end in the same state bit for bit, with the same pair count.
    NOTE: a CROSS-PARTITION check anchored at step 1 (test_full_size_sampled_parity_n262144 and bench.py's parity leg
    compare step 1 with the oracle; the oracle cannot run this horizon in test time)."""
    cfg = nb.stock_config(particleCount=262144, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    outs, pairs = [], []
    for world in (1, 8):
        grp = nb.StepperGroup(world, cfg=cfg)
        grp.upload(bodies)
        grp.step(1000)
        outs.append(grp.download())
        pairs.append(sum(r.stats().pairs for r in grp.ranks))
        grp.close()
    assert outs[1].numBodies == outs[0].numBodies < 262144       # coincident bodies do get deleted even at radius 0
    assert np.array_equal(bits(outs[1].block), bits(outs[0].block))
    assert pairs[0] == pairs[1]


@pytest.mark.parametrize("n", [100003, 50003], ids=["one-lane-kernel", "ring-kernel"])
def test_ragged_large_n_sampled(nb, n):
    """N = 100 003 / 50 003 (not multiples of 128: frozen tail, truncated last tile, wrapped cyclic tiles; the
    automatic kernel choice is the one-lane kernel for the first, the ring kernel for the second), one step,
    oracle on samples of bodies including the last active block and the frozen tail, then the whole state."""
    cfg = nb.stock_config(particleCount=n, minRadius=5.0, maxRadius=20.0)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg)
    st.upload(bodies)
    st.step(1)
    out = st.download()
    assert out.numBodies <= n
    blk = bodies.contiguousData
    n_active = (n // 128) * 128
    for lo in (0, 4093, n_active - 20, n - 12):
        hi = min(lo + 24, n)
        P, V, M, R, dl, _ = ol.port_range(blk, n, lo, hi, DT, 100000, 100000, GROWTH)
        keep = M != 0
        if out.numBodies == n:       # no deletion anywhere: indices unchanged
            assert np.array_equal(bits(out.Positions[lo:hi]), bits(P))
            assert np.array_equal(bits(out.Velocities[lo:hi]), bits(V))
            assert np.array_equal(bits(out.Masses[lo:hi]), bits(M))
        assert keep.all() or out.numBodies < n
    # the whole state against the full oracle step (about 1e10 pairs on the host cores)
    ref = blk.copy()
    n1, *_ = ol.port_step(ref, n, DT, 100000, 100000, GROWTH, want_events=False)
    assert_bodies_equal(out, ref, n1, "N=%d step 1" % n)
    assert st.stats().pairs == ol.port().oracle_pairs_per_step(n, ol.LITERAL)
    st.close()


@pytest.mark.parametrize("precision,n", [(0, 1048576), (1, 131072), (1, 1048576)])
def test_max_size_sampled(nb, precision, n):
    """configs[4] size: N = 1 Mi bodies in fp32 and in fp64 (C5 itself: 1.1e12 pairs per step, 48 MiB block,
    include/vec2.h:6-17 layout) and a smaller fp64 case; the oracle on spread samples of bodies, bit-exact, and the
    evaluated-pair count."""
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg, precision)
    st = nb.Stepper(cfg, precision=precision)
    st.upload(bodies)
    st.step(1)
    out = st.download()
    assert out.numBodies == n
    dt, gr = (float(DT), float(GROWTH)) if precision else (DT, GROWTH)
    for lo in (0, 333 * 128 + 77, n - 128 - 5, n - 8):
        P, V, M, R, dl, _ = ol.port_range(bodies.contiguousData, n, lo, lo + 8, dt, 100000, 100000, gr)
        assert np.array_equal(bits(out.Positions[lo:lo + 8]), bits(P))
        assert np.array_equal(bits(out.Velocities[lo:lo + 8]), bits(V))
    assert st.stats().pairs == ol.port().oracle_pairs_per_step(n, ol.LITERAL)
    st.close()


def test_c5_shape_eight_ranks_equals_one(nb):
    """BASELINE.json configs[4] (C5) as it is sharded: N = 1 048 576 bodies in fp64 range-partitioned over 8 ranks
    (131 072 own bodies each, the fp64 production kernel) for 2 steps against the single-rank run, whole state, bit
    for bit; fp64 has no reference at all, so the anchor is the fp64 oracle of test_max_size_sampled."""
    n = 1048576
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg, nb.F64)
    grp = nb.StepperGroup(8, cfg=cfg, precision=nb.F64)
    grp.upload(bodies)
    assert [r.own_range()[1] for r in grp.ranks] == [131072] * 8
    grp.step(2)
    one = nb.Stepper(cfg, precision=nb.F64)
    one.upload(bodies)
    one.step(2)
    a, b = grp.download(), one.download()
    assert a.numBodies == b.numBodies
    assert np.array_equal(bits(a.block), bits(b.block))
    assert sum(r.stats().pairs for r in grp.ranks) == one.stats().pairs
    grp.close(); one.close()


def test_exchange_follows_the_live_count(nb):
    """The per-step all-gather moves slots laid out for the LIVE bound of the body count, not for the capacity: the bound
    is the count kLag = 4 steps back (the host waits for that step's Meta, so every rank of an RCCL run derives the same
    layout), it only shrinks, and the bytes received add up to exactly that.  4 ranks on one GPU, a dense field whose
    count collapses; the group runs the same exchange / unpack code as RCCL contexts (peer-copy transport)."""
    world, n0, field, lag = 4, 4096, 12000, 4
    cfg = nb.stock_config(particleCount=n0, fieldWidth=field, fieldHeight=field)
    bodies = nb.init_bodies(cfg)
    one = nb.Stepper(cfg)
    one.upload(bodies)
    counts = []
    for _ in range(14):
        one.step(1)
        counts.append(one.body_count())
    assert counts[-1] < n0 - 512, "the case is meant to lose bodies: %r" % counts

    def stride(n):
        own_upper = ((n + 127) // 128 + world - 1) // world * 128
        return (32 + own_upper * 24 + 255) // 256 * 256

    grp = nb.StepperGroup(world, cfg=cfg)
    grp.upload(bodies)
    assert grp.ranks[1].stats().slot_bytes_now == stride(n0) and grp.ranks[1].stats().exchange_bytes == 0
    expect = 0
    for s in range(14):                                  # step s is laid out for the count after step s - lag
        grp.step(1)
        expect += world * stride(n0 if s < lag else counts[s - lag])
        st = grp.ranks[s % world].stats()
        assert st.exchange_bytes == expect, (s, st.exchange_bytes, expect)
    assert grp.ranks[0].stats().slot_bytes_now == stride(counts[13 - lag]) < stride(n0)
    a, b = grp.download(), one.download()
    assert a.numBodies == b.numBodies == counts[-1] and np.array_equal(bits(a.block), bits(b.block))
    # the download gathers velocities and Meta through the same interface: rank 0 received them
    assert grp.ranks[0].stats().exchange_bytes > expect and grp.ranks[1].stats().exchange_bytes == expect
    grp.close(); one.close()


def test_group_ranks_must_share_their_history(nb):
    """The ranks of a group lay their slots out from the same history of body counts: a rank that was uploaded again on its
    own is refused (NBODY_ERR_STATE) instead of exchanging slots of another layout."""
    cfg = nb.stock_config(particleCount=2048, fieldWidth=8000, fieldHeight=8000)
    bodies = nb.init_bodies(cfg)
    grp = nb.StepperGroup(2, cfg=cfg)
    grp.upload(bodies)
    grp.step(6)
    grp.ranks[1].upload(bodies)
    with pytest.raises(nb.NbodyError) as e:
        grp.step(1)
    assert e.value.status == -9 and "together" in str(e.value)
    grp.upload(bodies)
    grp.step(2)
    grp.close()


def test_group_context_alone_cannot_assemble_the_state(nb):
    """A rank of a group has its own velocities only: downloading it by itself returns the replica with zero velocities
    outside its range (documented), the group download returns the whole state."""
    cfg = nb.stock_config(particleCount=1024, fieldWidth=5000, fieldHeight=5000)
    bodies = nb.init_bodies(cfg)
    grp = nb.StepperGroup(2, cfg=cfg)
    grp.upload(bodies)
    grp.step(3)
    whole = grp.download()
    lo, cnt = grp.ranks[1].own_range()
    part = grp.ranks[1].download()
    assert part.numBodies == whole.numBodies
    assert np.array_equal(bits(part.Positions), bits(whole.Positions))
    assert np.array_equal(bits(part.Velocities[lo:lo + cnt]), bits(whole.Velocities[lo:lo + cnt]))
    assert not part.Velocities[:lo].any()
    grp.close()


def test_rccl_path_single_rank(nb):
    """The multi-rank code path on one GPU: RCCL loaded by dlopen, a 1-rank communicator from a unique id, the
    per-step slot all-gather and the all-gather based download.  (N>1 ranks cannot run on a 1-GPU box; the
    partition logic itself is covered by test_sharded_group_equals_single and the gloo CPU tests.)"""
    z = np.load(os.path.join(GOLD, "steps_dense_n1000.npz"))
    dt, growth, fw, fh = z["params"]
    st = nb.Stepper(capacity=1000, timestep=float(dt), growthRate=float(growth), fieldWidth=int(fw),
                    fieldHeight=int(fh), comm_id=nb.comm_unique_id(), force_comm=True)
    st.upload(nb.BodiesData.from_block(z["init"].view(np.float32), 1000))
    st.step(5)
    assert_bodies_equal(st.download(), z["after_5"].view(np.float32), int(z["counts"][4]), "rccl world=1")
    st.close()


def test_full_size_sampled_parity_n262144(nb):
    """BASELINE.json metric size: one step at N=262144 on the GPU, oracle on a spread sample of bodies."""
    n = 262144
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg)
    st.upload(bodies)
    st.step(1)
    out = st.download()
    assert out.numBodies == n
    for lo in (0, 127 * 128 + 100, n // 2 - 8, n - 16):
        P, V, M, R, dl, _ = ol.port_range(bodies.contiguousData, n, lo, lo + 16, DT, 100000, 100000, GROWTH)
        assert np.array_equal(bits(out.Positions[lo:lo + 16]), bits(P))
        assert np.array_equal(bits(out.Velocities[lo:lo + 16]), bits(V))
        assert np.array_equal(bits(out.Masses[lo:lo + 16]), bits(M))
    assert st.stats().pairs == ol.port().oracle_pairs_per_step(n, ol.LITERAL)
    st.close()


@pytest.mark.parametrize("general", [0, 1, 2], ids=["production-kernel", "general-kernel", "one-lane-kernel"])
@pytest.mark.parametrize("path", [p for p in STEP_FILES if "long_" not in p],
                         ids=[os.path.basename(p)[6:-4] for p in STEP_FILES if "long_" not in p])
def test_reference_shaped_launches(nb, path, general, monkeypatch):
    """nbody_launch_compute_forces_f32 / nbody_launch_move_bodies_f32 on a caller-owned device block in the
    reference layout (drop-in for src/nbody.cu:481-483), followed by the host compaction, on every golden case;
    device memory comes from torch.  With the reference's own block count the launch runs the production (ring) kernel
    through the launch workspace; NBODY_REF_LAUNCH_GENERAL=1 forces the general kernel that serves other block counts,
    NBODY_REF_LAUNCH_ONE_LANE=1 the one-lane-per-body kernel directly on the block layout."""
    import torch
    monkeypatch.setenv("NBODY_REF_LAUNCH_GENERAL", "1" if general == 1 else "0")
    monkeypatch.setenv("NBODY_REF_LAUNCH_ONE_LANE", "1" if general == 2 else "0")
    z = np.load(path)
    dt, growth, fw, fh = z["params"]
    n = int(z["n0"])
    host = z["init"].view(np.float32).copy()
    dev = torch.from_numpy(host.copy()).cuda()
    for s in range(1, min(len(z["counts"]), 12) + 1):
        upd_m = dev[4 * n:5 * n].clone()      # src/nbody.cu:467-470
        upd_r = dev[5 * n:6 * n].clone()
        blocks = nb.lib.nbody_num_blocks(n)
        stream = torch.cuda.current_stream().cuda_stream
        assert nb.lib.nbody_launch_compute_forces_f32(dev.data_ptr(), upd_m.data_ptr(), upd_r.data_ptr(), n,
                                                      float(dt), int(fw), int(fh), blocks, float(growth),
                                                      stream) == 0
        assert nb.lib.nbody_launch_move_bodies_f32(dev.data_ptr(), upd_m.data_ptr(), upd_r.data_ptr(), n,
                                                   float(dt), blocks, stream) == 0
        torch.cuda.synchronize()
        blk = dev.cpu().numpy()
        if s == 1:
            assert np.array_equal(bits(blk), z["pre_1"])
        n_new = nb.lib.nbody_block_compact(blk.ctypes.data, n, nb.F32)     # src/nbody.cu:488-510
        assert n_new == z["counts"][s - 1]
        if "after_%d" % s in z:
            assert np.array_equal(bits(blk[:6 * n_new]), z["after_%d" % s])
        n = n_new
        dev = torch.from_numpy(blk[:6 * n].copy()).cuda()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n,field,steps", [(1000, 5000, 6), (1024, 5000, 6), (130, 1500, 5), (77, 1000, 5),
                                           (4096, 100000, 2)])
def test_clean_semantics_matches_oracle(nb, n, field, steps, variant):
    """NBODY_CLEAN: every body active, all pairs, ascending j.  No reference exists for it (it is
    what the reference was meant to compute): checked bit-exactly against the oracle's clean mode."""
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg, semantics=nb.CLEAN, record_events=True, kernel_variant=variant)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    pairs = 0
    for s in range(steps):
        st.step(1)
        pairs += cur * (cur - 1)
        cur, stats, ab, de, _ = ol.port_step(blk, cur, DT, field, field, GROWTH, semantics=ol.CLEAN)
        ev = st.events()
        ev = ev[ev["step"] == s]
        assert sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0]) == \
            sorted((int(x), int(y)) for x, y in ab), "E_t step %d" % s
        assert_bodies_equal(st.download(), blk, cur, "clean step %d" % s)
    assert st.stats().pairs == pairs
    st.close()


def test_clean_semantics_sharded_and_fp64(nb):
    cfg = nb.stock_config(particleCount=900, fieldWidth=4000, fieldHeight=4000)
    for precision, dt, gr in ((nb.F32, DT, GROWTH), (nb.F64, float(DT), float(GROWTH))):
        bodies = nb.init_bodies(cfg, precision)
        grp = nb.StepperGroup(3, cfg=cfg, precision=precision, semantics=nb.CLEAN)
        grp.upload(bodies)
        grp.step(5)
        blk = bodies.contiguousData.copy()
        cur = 900
        for s in range(5):
            cur, *_ = ol.port_step(blk, cur, dt, 4000, 4000, gr, semantics=ol.CLEAN, want_events=False)
        out = grp.download()
        assert out.numBodies == cur and cur < 900
        assert np.array_equal(bits(out.block), bits(blk[:6 * cur]))
        grp.close()


def test_state_save_restore(nb, tmp_path):
    """Dump after 3 steps, restore into a fresh context, continue: same bits as the uninterrupted run; the dump's
    payload is the reference block layout of the survivors."""
    z = np.load(os.path.join(GOLD, "steps_dense_n1024.npz"))
    dt, growth, fw, fh = z["params"]
    mk = lambda: nb.Stepper(capacity=1024, timestep=float(dt), growthRate=float(growth), fieldWidth=int(fw),
                            fieldHeight=int(fh), record_events=True)
    a = mk()
    a.upload(nb.BodiesData.from_block(z["init"].view(np.float32), 1024))
    a.step(3)
    path = str(tmp_path / "state.bin")
    a.save_state(path)
    prec, n, steps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    assert nb.lib.nbody_state_peek(os.fsencode(path), ctypes.byref(prec), ctypes.byref(n), ctypes.byref(steps)) == 0
    assert (prec.value, n.value, steps.value) == (nb.F32, int(z["counts"][2]), 3)
    raw = np.fromfile(path, dtype=np.uint32, offset=64)
    assert np.array_equal(raw, bits(a.download().block))
    b = mk()
    b.load_state(path)
    a.step(2)
    b.step(2)
    assert_bodies_equal(b.download(), z["after_5"].view(np.float32), int(z["counts"][4]), "restored run")
    assert_bodies_equal(a.download(), z["after_5"].view(np.float32), int(z["counts"][4]), "uninterrupted run")
    assert set(b.events()["step"]) <= {3, 4}          # the step counter was restored too
    a.close(); b.close()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:     # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_distributed_control_path_one_rank():
    """bench.py exactly as the driver launches it for N>1 (torch.distributed.run, gloo rendezvous on 127.0.0.1,
    communicator id broadcast, RCCL slot all-gather, collective download, self-check against a single-rank run),
    rehearsed with ONE rank because the box has one GPU.  tests/test_gpu_multirank.py runs it with two ranks where
    two GPUs are visible."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--bodies", "16384", "--stock-radii", "--force-comm", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert d["parity_rccl_path"]["bitwise_equal"] is True, d["parity_rccl_path"]
    assert d["parity_rccl_path"]["bodies_after"] < 16384          # deletions happened: the slots were ragged
    assert abs(d["roofline"]["algorithmic_bytes_per_launch"] - 48 * 16384) <= 48 * 2000   # 48 B per body (fp32)
    assert 0 < d["roofline"]["valu"]["frac"] < 1


def test_cli_matches_oracle(nb, tmp_path):
    """The `nbody` command-line driver (C host code over the C ABI): reads ./nbodyConfig.txt from the current
    directory like the reference binary, echoes it, steps, and (--dump) emits the survivors' block."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ppa-nbody-collisions_amd", "nbody")
    cfg = nb.stock_config(particleCount=1500, totalIterations=7, fieldWidth=6000, fieldHeight=6000)
    nb.write_config(str(tmp_path / "nbodyConfig.txt"), cfg)
    r = subprocess.run([exe, "--dump"], cwd=str(tmp_path), capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    blk = nb.init_bodies(cfg).contiguousData.copy()
    n = 1500
    for s in range(7):
        n, *_ = ol.port_step(blk, n, DT, 6000, 6000, GROWTH, want_events=False)
    head, _, rest = out.partition(b"=====================\n")
    assert head.startswith(b"Running simulation with the following settings:\nparticleCount=1500\n")
    assert (b"Bodies left: %d\n" % n) in rest
    marker = rest.index(b"body-pair-interactions/sec\n") + len(b"body-pair-interactions/sec\n")
    payload = rest[marker:marker + 24 * n]
    assert np.array_equal(np.frombuffer(payload, dtype=np.uint32), blk[:6 * n].view(np.uint32))


@pytest.mark.parametrize("n,field,w,h", [(1024, 100000, 1024, 1024), (1000, 5000, 320, 200), (300, 2000, 64, 64)])
def test_render_image_matches_oracle(nb, tmp_path, n, field, w, h):
    """generateImage + saveImageToDisk: the raster of the state after each step equals the
    oracle's (which equals the reference's generateImage run through the shim, tests/test_oracle_cpu.py),
    including the reference's stale block count (bodies past 128*(N_before/128) are not drawn)."""
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field)
    bodies = nb.init_bodies(cfg)
    st = nb.Stepper(cfg)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(3):
        blocks = 1 if cur < 128 else cur // 128
        st.step(1)
        cur, *_ = ol.port_step(blk, cur, DT, field, field, GROWTH, want_events=False)
        assert np.array_equal(st.render_image(w, h), ol.port_render(blk, cur, blocks, w, h, field, field)), s
    path = str(tmp_path / "iteration_0.ppm")
    img = st.render_image(w, h)
    nb.saveImageToDisk(path, img)
    raw = open(path, "rb").read()
    head = b"P5\n%d %d\n255\n" % (w, h)
    assert raw.startswith(head) and raw[len(head):] == img.tobytes()
    st.close()


def test_render_image_matches_golden(nb):
    """The committed images come from the reference's own generateImage (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLD, "render_n300.npz"))
    n, field, w, h = [int(x) for x in z["params"]]
    st = _stepper(nb, n, field, field)
    st.upload(nb.BodiesData.from_block(z["init"].view(np.float32), n))
    for s in range(1, 4):
        st.step(1)
        assert np.array_equal(st.render_image(w, h), z["img_%d" % s]), s
    st.close()


def test_cli_images(nb, tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ppa-nbody-collisions_amd", "nbody")
    cfg = nb.stock_config(particleCount=640, totalIterations=7, save_Image_Every_Xth_Iteration=3, imgWidth=96,
                          imgHeight=80, fieldWidth=4000, fieldHeight=4000)
    nb.write_config(str(tmp_path / "nbodyConfig.txt"), cfg)
    r = subprocess.run([exe, "--images"], cwd=str(tmp_path), capture_output=True, timeout=120)
    assert r.returncode == 1 and b"Ensure the the folder exists" in r.stderr      # reference: exit(1), :365-370
    os.mkdir(str(tmp_path / "iter_img"))
    r = subprocess.run([exe, "--images"], cwd=str(tmp_path), capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert sorted(os.listdir(str(tmp_path / "iter_img"))) == ["iteration_0.ppm", "iteration_3.ppm"]   # 6 is last
    blk = nb.init_bodies(cfg).contiguousData.copy()
    cur = 640
    for k in range(4):
        blocks = 1 if cur < 128 else cur // 128
        cur, *_ = ol.port_step(blk, cur, DT, 4000, 4000, GROWTH, want_events=False)
        if k in (0, 3):
            raw = open(str(tmp_path / "iter_img" / ("iteration_%d.ppm" % k)), "rb").read()
            want = ol.port_render(blk, cur, blocks, 96, 80, 4000, 4000)
            assert raw == b"P5\n96 80\n255\n" + want.tobytes(), k


def test_error_paths(nb):
    """Every misuse returns an error code + message; nothing exits, throws across the ABI or corrupts state."""
    cfg = nb.stock_config(particleCount=256)
    with pytest.raises(nb.NbodyError) as e:
        nb.Stepper(cfg, capacity=0)
    assert e.value.status == -1
    with pytest.raises(nb.NbodyError) as e:
        nb.Stepper(cfg, device=99)
    assert e.value.status == -1
    with pytest.raises(nb.NbodyError) as e:
        nb.Stepper(cfg, world=2, rank=2)
    assert e.value.status == -1
    with pytest.raises(nb.NbodyError) as e:
        nb.Stepper(cfg, world=2, rank=0)            # RCCL context without a communicator id
    assert e.value.status == -1
    with pytest.raises(nb.NbodyError) as e:
        nb.Stepper(cfg, semantics=7)
    assert e.value.status == -1
    st = nb.Stepper(cfg, record_events=True, event_capacity=4)
    with pytest.raises(nb.NbodyError) as e:
        st.step(1)                                   # step before upload
    assert e.value.status == -9
    with pytest.raises(nb.NbodyError) as e:
        st.download()
    assert e.value.status == -9
    with pytest.raises(nb.NbodyError) as e:
        st.upload(nb.init_bodies(nb.stock_config(particleCount=257)))   # beyond capacity
    assert e.value.status == -7
    with pytest.raises(nb.NbodyError) as e:
        st.upload(nb.init_bodies(cfg, nb.F64)) if False else st.load_state("/nonexistent/state.bin")
    assert e.value.status == -2
    # event log overflow: counted, not stored, and reported through `total`
    dense = nb.stock_config(particleCount=256, fieldWidth=800, fieldHeight=800)
    st2 = nb.Stepper(dense, record_events=True, event_capacity=4)
    st2.upload(nb.init_bodies(dense))
    st2.step(1)
    buf = np.zeros(64, dtype=nb.EVENT_DTYPE)
    total = ctypes.c_int64(0)
    assert nb.lib.nbody_get_events(st2._ctx, buf.ctypes.data, 64, ctypes.byref(total)) == 0
    assert total.value > 4 and np.count_nonzero(buf["i"] | buf["j"]) <= 4
    # the context is still usable after all of that
    st.upload(nb.init_bodies(cfg))
    st.step(2)
    assert st.body_count() <= 256
    st.close(); st2.close()


@pytest.mark.parametrize("variant", [0, 1], ids=["production-kernel", "general-kernel"])
@pytest.mark.parametrize("n,field,max_r", [(1500, 6000, 200.0), (4096, 100000, 200.0), (5000, 100000, 0.0)])
def test_fp64_matches_oracle(nb, n, field, max_r, variant):
    """fp64 twin (configs[4] shape, small): both fp64 kernels (fast chain / compiler IEEE sqrt and divide) bit-exactly
    against the fp64 instantiation of the oracle, with events, dense and sparse, with and without radii.  (The reference
    has no fp64 code; what pins fp64 is its kernel text read at double precision: test_gpu_reference_kernels.py.)"""
    cfg = nb.stock_config(particleCount=n, fieldWidth=field, fieldHeight=field, minRadius=0.0, maxRadius=max_r)
    bodies = nb.init_bodies(cfg, nb.F64)
    st = nb.Stepper(cfg, precision=nb.F64, kernel_variant=variant, record_events=True)
    st.upload(bodies)
    blk = bodies.contiguousData.copy()
    cur = n
    for s in range(4):
        st.step(1)
        cur, stats, ab, de, _ = ol.port_step(blk, cur, float(DT), field, field, float(GROWTH))
        ev = st.events()
        ev = ev[ev["step"] == s]
        assert sorted((int(e["i"]), int(e["j"])) for e in ev[ev["kind"] == 0]) == \
            sorted((int(a), int(b)) for a, b in ab), "E_t step %d" % s
        out = st.download()
        assert out.numBodies == cur
        assert np.array_equal(bits(out.block), bits(blk[:6 * cur])), "fp64 step %d" % s
    st.close()


def test_fp64_extreme_values(nb):
    """fp64 bodies outside the fast chain's domain (huge / non-finite / coincident / denormally close): the
    per-tile coordinate bound and the per-pair flag must hand them to the general code."""
    n = 1024
    cfg = nb.stock_config(particleCount=n, minRadius=0.0, maxRadius=0.0)
    bodies = nb.init_bodies(cfg, nb.F64)
    P, M = bodies.Positions, bodies.Masses
    P[100] = [1e200, -3e250]           # beyond the 2^249 coordinate bound
    P[300] = [np.inf, 5.0]
    P[301] = [np.nan, 7.0]
    P[500] = P[499]                    # coincident: d2 == 0 <= 0 -> collision
    P[700] = [3e-200, 1e-201]
    P[701] = [3e-200 + 1e-215, 1e-201]  # d2 far below 2^-500
    P[900] = [1.0e-3, 0]
    P[901] = [1.0e-3 + 2.0e-19, 0]
    M[600] = np.inf
    M[601] = np.nan
    for variant in (0, 1):
        st = nb.Stepper(cfg, precision=nb.F64, kernel_variant=variant)
        st.upload(bodies)
        st.step(2)
        out = st.download()
        blk = bodies.contiguousData.copy()
        cur = n
        for s in range(2):
            cur, *_ = ol.port_step(blk, cur, float(DT), 100000, 100000, float(GROWTH), want_events=False)
        assert out.numBodies == cur
        assert _nan_aware_equal(out.block, blk[:6 * cur]), "variant %d" % variant
        st.close()
