#!/usr/bin/env python3
"""Generates the committed golden fixtures in tests/golden/ from the REFERENCE ITSELF.

Runs only where /root/reference exists (the dev container): it drives oracle/_ref/libnbody_ref.so, which is
the reference's own kernel text (src/nbody.cu:126-292), own RNG (include/jbutil.h:514-562) and own config
parser (include/nbodyConfig.h:22-227) compiled for the CPU by oracle/Makefile.  The reference ships no
tests or golden vectors of its own, so these files are what pins the oracle.

    make -C oracle ref && python tests/golden/make_golden.py

Outputs (all data: inputs + expected outputs, no reference text):
    rng_kat.json            raw generator outputs
    init_stock.npz          first bodies of the stock initial condition
    config_cases.json       config texts (written by us) with the values/echo/exit status the reference gives
    steps_<case>.npz        initial block, per-step survivor counts, selected post-step blocks (raw fp32 bits)
    big_n65536.json         sha256 + sampled bodies of one literal step at N=65536 (stock radii and radii 0)
    render_n300.npz         images the reference's generateImage produces for a small dense case
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

DT = np.float32(0.2)
GROWTH = np.float32(0.1)


def rng_kat():
    L = ol.ref()
    out = {}
    for seed in (1024, 0, 1, 0xDEADBEEF12345678):
        iv = np.zeros(8, dtype=np.uint64)
        L.ref_rng_ival64(seed, 8, iv.ctypes.data)
        fv = np.zeros(8, dtype=np.float64)
        L.ref_rng_fval(seed, 8, -3.5, 1e17, fv.ctypes.data)
        out[str(seed)] = {"ival64": ["%016x" % int(x) for x in iv],
                          "fval_m3p5_1e17_bits": ["%016x" % int(x) for x in fv.view(np.uint64)]}
    json.dump(out, open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)


def init_stock():
    b = ol.ref_init(64)
    b2 = ol.ref_init(48, 5000, 7000, 1.0, 1e6, 0.0, 0.0)
    np.savez(os.path.join(HERE, "init_stock.npz"), stock_n64=b.view(np.uint32), small_n48=b2.view(np.uint32))


CONFIG_CASES = {
    "stock": None,  # filled from the reference's own nbodyConfig.txt bytes at generation time
    "suffixes": "particleCount=1024abc\ntimestep=  0.25f\nradiusGrowthRate=1e-1f\nminRandBodyMass=0x1p4\n"
                "maxRandBodyMass=1E17\nminRadius=.5\nmaxRadius=200.f\nfieldWidth=-5\nfieldHeight=+7\n",
    "unknown_and_blank": "foo=1\n\nparticleCount=12\n=5\nnoequals\nimagePath=a=b=c d\nparticleCount =3\n",
    "crlf": "particleCount=77\r\ntimestep=0.5\r\nimagePath=out\r\n",
    "no_trailing_newline": "particleCount=5\ntotalIterations=9",
    "duplicate_keys": "particleCount=5\nparticleCount=6\n",
    "bad_int": "totalIterations=abc\nparticleCount=4\n",
    "bad_float": "timestep=f0.2\n",
    "int_overflow": "imgWidth=99999999999\n",
    "float_overflow": "maxRandBodyMass=1e60\n",
    "key_without_equals": "particleCount\n",
    "empty": "",
    "float_as_int": "particleCount=12.9\nimgHeight=-3\nsave_Image_Every_Xth_Iteration= 10\n",
}

_PARSE_CHILD = r"""
import ctypes, sys, json
class C(ctypes.Structure):
    _fields_=[(k,ctypes.c_int) for k in ("particleCount","totalIterations","saveEvery")]+\
             [(k,ctypes.c_float) for k in ("timestep","minMass","maxMass","minRadius","maxRadius","growthRate")]+\
             [(k,ctypes.c_int) for k in ("imgWidth","imgHeight","fieldWidth","fieldHeight")]+[("imagePath",ctypes.c_char*256)]
L=ctypes.CDLL(sys.argv[1]); c=C()
L.ref_parse_config(sys.argv[2].encode(), ctypes.byref(c))
sys.stdout.flush()
d={k:getattr(c,k) for k,_ in C._fields_ if k!="imagePath"}
d["imagePath"]=c.imagePath.decode("latin-1")
import struct
for k in ("timestep","minMass","maxMass","minRadius","maxRadius","growthRate"):
    d[k]="%08x"%struct.unpack("<I",struct.pack("<f",d[k]))[0]
sys.stderr.write(json.dumps(d))
"""

FILE_KEY_TO_FIELD = {"particleCount": "particleCount", "totalIterations": "totalIterations",
                     "save_Image_Every_Xth_Iteration": "saveEvery", "timestep": "timestep",
                     "minRandBodyMass": "minMass", "maxRandBodyMass": "maxMass", "minRadius": "minRadius",
                     "maxRadius": "maxRadius", "radiusGrowthRate": "growthRate", "imgWidth": "imgWidth",
                     "imgHeight": "imgHeight", "fieldWidth": "fieldWidth", "fieldHeight": "fieldHeight",
                     "imagePath": "imagePath"}


def config_cases():
    cases = dict(CONFIG_CASES)
    cases["stock"] = open("/root/reference/nbodyConfig.txt", "rb").read().decode("latin-1")
    out = {}
    for name, text in cases.items():
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "nbodyConfig.txt")
            open(p, "wb").write(text.encode("latin-1"))
            r = subprocess.run([sys.executable, "-c", _PARSE_CHILD, ol.REF_SO, p], capture_output=True)
            entry = {"text": text, "exit": r.returncode, "echo": r.stdout.decode("latin-1")}
            if r.returncode == 0:
                vals = json.loads(r.stderr.decode())
                # keep only the keys this file sets successfully (others are uninitialised in the reference)
                keys_set = []
                for line in text.split("\n"):
                    k = line.split("=", 1)[0]
                    if k in FILE_KEY_TO_FIELD and k not in keys_set:
                        keys_set.append(k)
                entry["values"] = {k: vals[FILE_KEY_TO_FIELD[k]] for k in keys_set}
            out[name] = entry
    # missing file
    r = subprocess.run([sys.executable, "-c", _PARSE_CHILD, ol.REF_SO, "/nonexistent/nbodyConfig.txt"],
                       capture_output=True)
    out["__missing_file__"] = {"text": None, "exit": r.returncode, "echo": r.stdout.decode("latin-1")}
    json.dump(out, open(os.path.join(HERE, "config_cases.json"), "w"), indent=1)


def run_case(name, block, n, steps, keep, fw=100000, fh=100000, dt=DT, growth=GROWTH, keep_pre=(1,)):
    """Steps the literal oracle; stores the initial block, survivor counts, blocks after the steps in `keep`
    and the pre-compaction block of the steps in `keep_pre` (1-based step numbers)."""
    data = {"init": block[:6 * n].view(np.uint32).copy(), "n0": np.int32(n),
            "params": np.array([float(dt), float(growth), fw, fh], dtype=np.float64)}
    counts = []
    b = block.copy()
    for s in range(1, steps + 1):
        n, pre = ol.ref_step(b, n, dt, fw, fh, growth, pre=s in keep_pre)
        counts.append(n)
        if s in keep:
            data["after_%d" % s] = b[:6 * n].view(np.uint32).copy()
        if s in keep_pre:
            data["pre_%d" % s] = pre.view(np.uint32).copy()
    data["counts"] = np.array(counts, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "steps_%s.npz" % name), **data)
    print(name, "final n", n)


def step_cases():
    # the commented-out hand scenario of src/nbody.cu:418-429
    b = ol.make_block([[-500, 0], [500, 0], [-600, -150]], [[10, 0], [-10, 0], [0, 0]], [1e10, 1e14, 1e3],
                      [10, 20, 7])
    run_case("three_body", b, 3, 3, keep=(1, 2, 3))
    # C1: stock config at N=1024, 100 steps (BASELINE.json configs[0])
    run_case("c1_n1024", ol.ref_init(1024), 1024, 100, keep=(1, 2, 100))
    # edge cases of the index semantics: N<128, N=128, 129 (no interactions), 130, 255, 257
    for n in (1, 2, 100, 127, 128, 129, 130, 200, 255, 257):
        run_case("edge_n%d" % n, ol.ref_init(n, 3000, 3000), n, 12, keep=(1, 12), fw=3000, fh=3000)
    # ragged start + dense field: many collisions, frozen tail bodies (quirk Q2), shrinking N
    run_case("dense_n1000", ol.ref_init(1000, 5000, 5000), 1000, 40, keep=(1, 5, 40), fw=5000, fh=5000)
    run_case("dense_n1024", ol.ref_init(1024, 5000, 5000), 1024, 40, keep=(1, 5, 40), fw=5000, fh=5000)
    # no-collision configuration (radii 0): C2 shape at a size the fixture can hold
    run_case("r0_n2048", ol.ref_init(2048, min_r=0.0, max_r=0.0), 2048, 10, keep=(1, 10))
    run_case("stock_n4096", ol.ref_init(4096), 4096, 6, keep=(1, 6))
    # the north_star's horizon: 1000 steps (C2/C3 length) at a size the reference shim finishes in a minute,
    # stock field and radii: bodies keep merging all the way through (2048 -> 741)
    run_case("long_n2048", ol.ref_init(2048, 100000, 100000), 2048, 1000, keep=(1, 100, 500, 1000))


def big_cases():
    out = {}
    for name, kw in (("stock_radii", {}), ("radii0", {"min_r": 0.0, "max_r": 0.0})):
        n = 65536
        b = ol.ref_init(n, **kw)
        n1, pre = ol.ref_step(b, n, DT, 100000, 100000, GROWTH, pre=True)
        idx = [0, 1, 2, 127, 128, 4095, 32768, 65407, 65408, 65535]
        u = pre.view(np.uint32)
        out[name] = {"n0": n, "n1": int(n1), "sha256_pre": hashlib.sha256(pre.tobytes()).hexdigest(),
                     "sha256_post": hashlib.sha256(b[:6 * n1].tobytes()).hexdigest(),
                     "sample_idx": idx,
                     "sample_pre": [["%08x" % int(x) for x in (u[2 * i], u[2 * i + 1], u[2 * n + 2 * i],
                                                             u[2 * n + 2 * i + 1], u[4 * n + i], u[5 * n + i])]
                                    for i in idx]}
        print("big", name, n1)
    json.dump(out, open(os.path.join(HERE, "big_n65536.json"), "w"), indent=1)


def render_cases():
    """generateImage (src/nbody.cu:294-348) of the dense N=300 state after each of 3 steps, 64x48 pixels."""
    n, field, w, h = 300, 2000, 64, 48
    b = ol.ref_init(n, field, field)
    out = {"init": b.view(np.uint32).copy(), "params": np.array([n, field, w, h], dtype=np.int64)}
    cur = n
    for s in range(1, 4):
        blocks = 1 if cur < 128 else cur // 128
        cur, _ = ol.ref_step(b, cur, DT, field, field, GROWTH)
        out["img_%d" % s] = ol.ref_render(b, cur, blocks, w, h, field, field)
        out["n_%d" % s] = np.int32(cur)
    np.savez_compressed(os.path.join(HERE, "render_n300.npz"), **out)


if __name__ == "__main__":
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle ref"
    rng_kat()
    init_stock()
    config_cases()
    step_cases()
    big_cases()
    render_cases()
