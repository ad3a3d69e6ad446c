"""ctypes access to the CHECKERS (test infrastructure): oracle/_build/libnbody_oracle.so (our C restatement)
and oracle/_ref/libnbody_ref.so (the reference's own kernel text behind a CPU shim, present only where it
was built from /root/reference).  Nothing in the product package imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT_SO = os.path.join(ROOT, "oracle", "_build", "libnbody_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref.so")
REF_FMA_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref_fma.so")   # g++ -ffp-contract=fast -mfma build

LITERAL, CLEAN = 0, 1


class OracleStats(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_int64), ("n_absorb", ctypes.c_int32), ("n_deleted", ctypes.c_int32),
                ("n_active", ctypes.c_int32), ("n_after", ctypes.c_int32)]


def build_port():
    if not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SO) < max(
            os.path.getmtime(os.path.join(ROOT, "oracle", f))
            for f in ("nbody_oracle.c", "nbody_oracle_step.inc", "nbody_oracle.h")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"],
                              stdout=subprocess.DEVNULL)
    return PORT_SO


_port = None
_ref = None


def port():
    global _port
    if _port is None:
        L = ctypes.CDLL(build_port())
        vp, ip = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)
        for name, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            f = getattr(L, "oracle_step_" + name)
            f.argtypes = [vp, ip, real, ctypes.c_int, ctypes.c_int, real, ctypes.c_int,
                          vp, ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(OracleStats), vp]
            f.restype = ctypes.c_int
            g = getattr(L, "oracle_range_" + name)
            g.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, real, ctypes.c_int, ctypes.c_int,
                          real, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.POINTER(OracleStats)]
            g.restype = ctypes.c_int
        L.oracle_pairs_per_step.argtypes = [ctypes.c_int, ctypes.c_int]
        L.oracle_pairs_per_step.restype = ctypes.c_int64
        L.oracle_jlist.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        L.oracle_jlist.restype = ctypes.c_int
        L.oracle_render_f32.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int]
        L.oracle_render_f32.restype = None
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_get_max_threads.restype = ctypes.c_int
        _port = L
    return _port


def have_ref():
    return os.path.exists(REF_SO)


REF_HIP_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref_hip.so")          # the reference's kernels, hipcc, no contraction
REF_HIP_FMA_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref_hip_fma.so")  # ... hipcc's default contraction
_ref_hip = {}


def have_ref_hip():
    return os.path.exists(REF_HIP_SO) and os.path.exists(REF_HIP_FMA_SO)


REF_HIP_F64_SO = os.path.join(ROOT, "oracle", "_ref", "libnbody_ref_hip_f64.so")  # ... `float` read as `double`, no contraction


def have_ref_hip_f64():
    return os.path.exists(REF_HIP_F64_SO)


def ref_hip_run(block, n, steps, dt, fw, fh, growth, fma=False, pre=False):
    """`steps` iterations of the reference's own kernels ON THE GPU (oracle/ref_hip).  block: float32[>= 6n] - or float64
    for the fp64 reading of the reference's text (libnbody_ref_hip_f64.so) -, updated in place.
    Returns (new_n, kernel_ms_total, pre_compaction_block_of_last_step | None)."""
    f64 = block.dtype == np.float64
    assert not (f64 and fma)
    path = REF_HIP_F64_SO if f64 else (REF_HIP_FMA_SO if fma else REF_HIP_SO)
    real = ctypes.c_double if f64 else ctypes.c_float
    if path not in _ref_hip:
        L = ctypes.CDLL(path)
        L.refhip_run.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, real,
                                 ctypes.c_int, ctypes.c_int, real, ctypes.c_void_p,
                                 ctypes.POINTER(ctypes.c_double)]
        L.refhip_last_error.restype = ctypes.c_char_p
        assert L.refhip_real_bytes() == (8 if f64 else 4)
        _ref_hip[path] = L
    L = _ref_hip[path]
    cn = ctypes.c_int(n)
    ms = ctypes.c_double(0.0)
    preb = np.empty(6 * n, dtype=block.dtype) if pre else None
    rc = L.refhip_run(block.ctypes.data, ctypes.byref(cn), steps, dt, fw, fh, growth,
                      preb.ctypes.data if pre else None, ctypes.byref(ms))
    if rc != 0:
        raise RuntimeError("refhip_run failed (%d): %s" % (rc, L.refhip_last_error().decode()))
    return cn.value, ms.value, preb


_ref_fma = None


def ref_fma():
    """The FMA-contracted build of the literal shim (fixture generation only)."""
    global _ref_fma
    if _ref_fma is None:
        L = ctypes.CDLL(REF_FMA_SO)
        L.ref_step.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_float, ctypes.c_int,
                               ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _ref_fma = L
    return _ref_fma


def ref_fma_step(block, n, dt, fw, fh, growth, pre=False):
    cn = ctypes.c_int(n)
    preb = np.empty(6 * n, dtype=np.float32) if pre else None
    ref_fma().ref_step(block.ctypes.data, ctypes.byref(cn), dt, fw, fh, growth, preb.ctypes.data if pre else None)
    return cn.value, preb


def ref():
    global _ref
    if _ref is None:
        L = ctypes.CDLL(REF_SO)
        L.ref_step.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_float, ctypes.c_int,
                               ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        L.ref_init_bodies.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_float] * 4
        L.ref_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.ref_rng_ival64.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        L.ref_rng_fval.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_void_p]
        _ref = L
    return _ref


# ---------------------------------------------------------------------------------------------------------
# numpy helpers on the reference block layout [P | V | M | R]
# ---------------------------------------------------------------------------------------------------------
def carve(block, n):
    """Views (P[n,2], V[n,2], M[n], R[n]) into a flat real array holding >= 6n elements."""
    return (block[:2 * n].reshape(n, 2), block[2 * n:4 * n].reshape(n, 2), block[4 * n:5 * n],
            block[5 * n:6 * n])


def make_block(P, V, M, R, dtype=np.float32):
    n = len(M)
    b = np.empty(6 * n, dtype=dtype)
    p, v, m, r = carve(b, n)
    p[:] = P
    v[:] = V
    m[:] = M
    r[:] = R
    return b


def port_step(block, n, dt, fw, fh, growth, semantics=LITERAL, want_events=True, pre=False):
    """One oracle step in place. Returns (new_n, stats, absorb_pairs[k,2], deleted[k], pre_block|None)."""
    L = port()
    f64 = block.dtype == np.float64
    fn = L.oracle_step_f64 if f64 else L.oracle_step_f32
    cn = ctypes.c_int(n)
    st = OracleStats()
    cap = max(16, 4 * n) if want_events else 0
    ab = np.zeros((cap, 2), dtype=np.int32)
    de = np.zeros(max(cap, 1), dtype=np.int32)
    preb = np.empty(6 * n, dtype=block.dtype) if pre else None
    rc = fn(block.ctypes.data, ctypes.byref(cn), dt, fw, fh, growth, semantics,
            ab.ctypes.data if want_events else None, cap, de.ctypes.data if want_events else None, cap,
            ctypes.byref(st), preb.ctypes.data if pre else None)
    assert rc == 0
    return cn.value, st, ab[:st.n_absorb], de[:st.n_deleted], preb


def port_range(block, n, lo, hi, dt, fw, fh, growth, semantics=LITERAL):
    L = port()
    f64 = block.dtype == np.float64
    fn = L.oracle_range_f64 if f64 else L.oracle_range_f32
    q = hi - lo
    oP = np.empty((q, 2), block.dtype)
    oV = np.empty((q, 2), block.dtype)
    oM = np.empty(q, block.dtype)
    oR = np.empty(q, block.dtype)
    dl = np.zeros(max(q, 1), np.uint8)
    st = OracleStats()
    rc = fn(block.ctypes.data, n, lo, hi, dt, fw, fh, growth, semantics, oP.ctypes.data, oV.ctypes.data,
            oM.ctypes.data, oR.ctypes.data, dl.ctypes.data, ctypes.byref(st))
    assert rc == 0
    return oP, oV, oM, oR, dl[:q], st


def ref_step(block, n, dt, fw, fh, growth, pre=False):
    L = ref()
    cn = ctypes.c_int(n)
    preb = np.empty(6 * n, dtype=np.float32) if pre else None
    L.ref_step(block.ctypes.data, ctypes.byref(cn), dt, fw, fh, growth, preb.ctypes.data if pre else None)
    return cn.value, preb


def port_render(block, n, blocks, w, h, fw, fh):
    img = np.zeros(w * h, np.uint8)
    port().oracle_render_f32(block.ctypes.data, n, blocks, img.ctypes.data, w, h, fw, fh)
    return img.reshape(h, w)


def ref_render(block, n, blocks, w, h, fw, fh):
    img = np.zeros(w * h, np.uint8)
    ref().ref_render(block.ctypes.data, n, blocks, img.ctypes.data, w, h, fw, fh)
    return img.reshape(h, w)


def ref_init(n, fw=100000, fh=100000, min_mass=1e4, max_mass=1e17, min_r=50.0, max_r=200.0):
    b = np.zeros(6 * n, dtype=np.float32)
    ref().ref_init_bodies(b.ctypes.data, n, fw, fh, min_mass, max_mass, min_r, max_r)
    return b
