"""Python host side of the MI355X-native N-body + collision stepper.

Thin ctypes mirror of the C ABI in include/nbody.h (library: libnbody_mi355x.so next to this file).  The
class/function names follow the reference's own vocabulary (ConfigData / parseConfigFile /
BodiesData of /root/reference/include/nbodyConfig.h:4-19,22 and src/nbody.cu:47-124) so that tests read
like the reference's main() (src/nbody.cu:373-551).

There is NO fallback: if the shared library is missing the import fails, and every compute call fails with
NbodyError when no gfx950 device is visible.  PyTorch is not needed by this module.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnbody_mi355x.so")

F32, F64 = 0, 1
LITERAL, CLEAN = 0, 1
FLAG_RECORD_EVENTS = 1
FLAG_GROUP_EXCHANGE = 2
FLAG_FORCE_COMM = 4
COMM_ID_BYTES = 128
IMAGE_PATH_MAX = 1024

KEYS = ("particleCount", "totalIterations", "save_Image_Every_Xth_Iteration", "timestep", "minRandBodyMass",
        "maxRandBodyMass", "minRadius", "maxRadius", "radiusGrowthRate", "imgWidth", "imgHeight", "fieldWidth",
        "fieldHeight", "imagePath")


class NbodyError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (%d): %s" % (_status_name(status), status, message))
        self.status = status


class ConfigData(ctypes.Structure):
    """struct ConfigData, include/nbodyConfig.h:4-19 (+ `present` bitmask, see include/nbody.h)."""
    _fields_ = [("particleCount", ctypes.c_int), ("totalIterations", ctypes.c_int),
                ("save_Image_Every_Xth_Iteration", ctypes.c_int), ("timestep", ctypes.c_float),
                ("minRandBodyMass", ctypes.c_float), ("maxRandBodyMass", ctypes.c_float),
                ("minRadius", ctypes.c_float), ("maxRadius", ctypes.c_float), ("growthRate", ctypes.c_float),
                ("imgWidth", ctypes.c_int), ("imgHeight", ctypes.c_int), ("fieldWidth", ctypes.c_int),
                ("fieldHeight", ctypes.c_int), ("_imagePath", ctypes.c_char * IMAGE_PATH_MAX),
                ("present", ctypes.c_uint32)]

    @property
    def imagePath(self):
        return self._imagePath.decode("utf-8", "surrogateescape")

    def has(self, key):
        return bool(self.present >> KEYS.index(key) & 1)


class _CtxDesc(ctypes.Structure):
    _fields_ = [("precision", ctypes.c_int), ("semantics", ctypes.c_int), ("capacity", ctypes.c_int),
                ("device", ctypes.c_int), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("flags", ctypes.c_uint32), ("event_capacity", ctypes.c_int), ("timestep", ctypes.c_double),
                ("growthRate", ctypes.c_double), ("fieldWidth", ctypes.c_int), ("fieldHeight", ctypes.c_int),
                ("comm_id", ctypes.c_void_p), ("kernel_variant", ctypes.c_int)]


class Stats(ctypes.Structure):
    _fields_ = [("steps", ctypes.c_int64), ("pairs", ctypes.c_int64), ("force_kernel_ms", ctypes.c_double),
                ("force_kernel_launches", ctypes.c_int64), ("n_bodies", ctypes.c_int), ("n_own", ctypes.c_int),
                ("exchange_ms", ctypes.c_double), ("exchange_launches", ctypes.c_int64),
                ("exchange_bytes", ctypes.c_int64), ("slot_bytes_now", ctypes.c_int64)]


class Rng(ctypes.Structure):
    _fields_ = [("u", ctypes.c_uint64), ("v", ctypes.c_uint64), ("w", ctypes.c_uint64)]


EVENT_DTYPE = np.dtype([("step", np.int32), ("i", np.int32), ("j", np.int32), ("kind", np.int32)])

# every symbol include/nbody.h declares: name -> (restype, argtypes)
_vp, _ip, _i, _f, _d, _sz = (ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float,
                             ctypes.c_double, ctypes.c_size_t)
_pp = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "nbody_last_error_string": (ctypes.c_char_p, []),
    "nbody_status_string": (ctypes.c_char_p, [_i]),
    "nbody_abi_version": (_i, []),
    "nbody_block_bytes": (_sz, [_i, _i]),
    "nbody_block_alloc": (_vp, [_i, _i]),
    "nbody_block_free": (None, [_vp]),
    "nbody_block_carve_f32": (_i, [_vp, _i, _pp, _pp, _pp, _pp]),
    "nbody_block_carve_f64": (_i, [_vp, _i, _pp, _pp, _pp, _pp]),
    "nbody_block_compact": (_i, [_vp, _i, _i]),
    "nbody_config_parse": (_i, [ctypes.c_char_p, ctypes.POINTER(ConfigData)]),
    "nbody_config_parse_fd": (_i, [ctypes.c_char_p, ctypes.POINTER(ConfigData), _i]),
    "nbody_config_stock": (None, [ctypes.POINTER(ConfigData)]),
    "nbody_rng_seed": (None, [ctypes.POINTER(Rng), ctypes.c_uint64]),
    "nbody_rng_ival64": (ctypes.c_uint64, [ctypes.POINTER(Rng)]),
    "nbody_rng_fval": (_d, [ctypes.POINTER(Rng)]),
    "nbody_rng_fval_range": (_d, [ctypes.POINTER(Rng), _d, _d]),
    "nbody_init_bodies": (_i, [ctypes.POINTER(ConfigData), _vp, _i]),
    "nbody_ctx_desc_from_config": (None, [ctypes.POINTER(_CtxDesc), ctypes.POINTER(ConfigData), _i]),
    "nbody_ctx_create": (_i, [_pp, ctypes.POINTER(_CtxDesc)]),
    "nbody_ctx_destroy": (_i, [_vp]),
    "nbody_upload": (_i, [_vp, _vp, _i]),
    "nbody_step": (_i, [_vp, _i]),
    "nbody_download": (_i, [_vp, _vp, _ip]),
    "nbody_body_count": (_i, [_vp, _ip]),
    "nbody_sync": (_i, [_vp]),
    "nbody_get_events": (_i, [_vp, _vp, _i, ctypes.POINTER(ctypes.c_int64)]),
    "nbody_clear_events": (_i, [_vp]),
    "nbody_get_stats": (_i, [_vp, ctypes.POINTER(Stats)]),
    "nbody_set_kernel_timing": (_i, [_vp, _i]),
    "nbody_force_kernel_name": (ctypes.c_char_p, [_vp]),
    "nbody_render_image": (_i, [_vp, _vp, _i, _i]),
    "nbody_write_pgm": (_i, [ctypes.c_char_p, _vp, _i, _i]),
    "nbody_ctx_info": (_i, [_vp, ctypes.POINTER(_CtxDesc), ctypes.POINTER(ctypes.c_int64)]),
    "nbody_ctx_set_steps": (_i, [_vp, ctypes.c_int64]),
    "nbody_state_save": (_i, [_vp, ctypes.c_char_p]),
    "nbody_state_load": (_i, [_vp, ctypes.c_char_p]),
    "nbody_state_peek": (_i, [ctypes.c_char_p, _ip, _ip, ctypes.POINTER(ctypes.c_int64)]),
    "nbody_comm_unique_id": (_i, [_vp]),
    "nbody_group_step": (_i, [_pp, _i, _i]),
    "nbody_group_download": (_i, [_pp, _i, _vp, _ip]),
    "nbody_own_range": (_i, [_vp, _ip, _ip]),
    "nbody_partition": (_i, [_i, _i, _i, _ip, _ip]),
    "nbody_ctx_stream": (_vp, [_vp]),
    "nbody_num_blocks": (_i, [_i]),
    "nbody_launch_compute_forces_f32": (_i, [_vp, _vp, _vp, _i, _f, _i, _i, _i, _f, _vp]),
    "nbody_launch_move_bodies_f32": (_i, [_vp, _vp, _vp, _i, _f, _i, _vp]),
    "nbody_launch_workspace_release": (_i, []),
    "nbody_selftest_ieee_f32": (_i, [_i, ctypes.POINTER(ctypes.c_uint64 * 3)]),
    "nbody_selftest_chain_f64": (_i, [_i, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64 * 2)]),
    "nbody_selftest_rcp_ones_f64": (_i, [_i, ctypes.POINTER(ctypes.c_uint64 * 5)]),
    "nbody_selftest_lds_record": (_i, [_i, _i, ctypes.POINTER(ctypes.c_uint64 * 3)]),
    "nbody_debug_ring_probe": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64 * 8)]),
    "nbody_debug_force_only": (_i, [_vp, _i]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C ppa-nbody-collisions_amd/csrc). There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _status_name(status):
    return lib.nbody_status_string(status).decode()


def _check(status):
    if status != 0:
        raise NbodyError(status, lib.nbody_last_error_string().decode())


# ---------------------------------------------------------------------------------------------------------
# config / bodies (host side)
# ---------------------------------------------------------------------------------------------------------
def parseConfigFile(path, echo=True):
    """parseConfigFile, include/nbodyConfig.h:22-227. Echoes like the reference unless echo=False.
    Raises NbodyError(NBODY_ERR_IO / NBODY_ERR_PARSE) where the reference calls exit(1)."""
    cfg = ConfigData()
    if echo:
        sys.stdout.flush()
        _check(lib.nbody_config_parse(os.fsencode(path), ctypes.byref(cfg)))
    else:
        _check(lib.nbody_config_parse_fd(os.fsencode(path), ctypes.byref(cfg), -1))
    return cfg


def stock_config(**overrides):
    """The stock nbodyConfig.txt (nbodyConfig.txt:1-14) with keyword overrides of struct fields."""
    cfg = ConfigData()
    lib.nbody_config_stock(ctypes.byref(cfg))
    for k, v in overrides.items():
        if k == "radiusGrowthRate":
            k = "growthRate"
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def write_config(path, cfg):
    """Writes cfg in the reference's nbodyConfig.txt format (key order of nbodyConfig.txt:1-14)."""
    with open(path, "w") as f:
        f.write("particleCount=%d\ntotalIterations=%d\nsave_Image_Every_Xth_Iteration=%d\n" %
                (cfg.particleCount, cfg.totalIterations, cfg.save_Image_Every_Xth_Iteration))
        f.write("timestep=%.9g\nradiusGrowthRate=%.9g\nminRandBodyMass=%.9g\nmaxRandBodyMass=%.9g\n" %
                (cfg.timestep, cfg.growthRate, cfg.minRandBodyMass, cfg.maxRandBodyMass))
        f.write("minRadius=%.9g\nmaxRadius=%.9g\nimgWidth=%d\nimgHeight=%d\nfieldWidth=%d\nfieldHeight=%d\n" %
                (cfg.minRadius, cfg.maxRadius, cfg.imgWidth, cfg.imgHeight, cfg.fieldWidth, cfg.fieldHeight))
        f.write("imagePath=%s\n" % cfg.imagePath)


class BodiesData:
    """Host body container with the reference's single-allocation layout (struct BodiesData,
    src/nbody.cu:47-124): one flat array [Positions | Velocities | Masses | Radii]."""

    def __init__(self, numBodies, precision=F32, capacity=None):
        self.precision = precision
        self.dtype = np.float64 if precision == F64 else np.float32
        self.capacity = max(int(capacity if capacity is not None else numBodies), int(numBodies), 1)
        self.contiguousData = np.zeros(6 * self.capacity, dtype=self.dtype)
        self.numBodies = int(numBodies)

    @classmethod
    def from_arrays(cls, P, V, M, R, precision=F32):
        b = cls(len(M), precision)
        b.Positions[:] = P
        b.Velocities[:] = V
        b.Masses[:] = M
        b.Radii[:] = R
        return b

    @classmethod
    def from_block(cls, block, numBodies, precision=F32):
        b = cls(numBodies, precision)
        b.contiguousData[:6 * numBodies] = np.asarray(block)[:6 * numBodies]
        return b

    # carving of src/nbody.cu:74-77 for the CURRENT numBodies
    @property
    def Positions(self):
        n = self.numBodies
        return self.contiguousData[:2 * n].reshape(n, 2)

    @property
    def Velocities(self):
        n = self.numBodies
        return self.contiguousData[2 * n:4 * n].reshape(n, 2)

    @property
    def Masses(self):
        n = self.numBodies
        return self.contiguousData[4 * n:5 * n]

    @property
    def Radii(self):
        n = self.numBodies
        return self.contiguousData[5 * n:6 * n]

    @property
    def block(self):
        return self.contiguousData[:6 * self.numBodies]

    @property
    def ptr(self):
        return self.contiguousData.ctypes.data

    def copy(self):
        b = BodiesData(self.numBodies, self.precision, self.capacity)
        b.contiguousData[:] = self.contiguousData
        return b


def init_bodies(cfg, precision=F32):
    """The initial-condition loop of src/nbody.cu:401-416 (seed 1024; x, y, m, r per body; v = 0)."""
    b = BodiesData(cfg.particleCount, precision)
    _check(lib.nbody_init_bodies(ctypes.byref(cfg), b.ptr, precision))
    return b


# ---------------------------------------------------------------------------------------------------------
# device stepper
# ---------------------------------------------------------------------------------------------------------
def saveImageToDisk(filename, img):
    """saveImageToDisk, src/nbody.cu:350-371 (binary PGM)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    _check(lib.nbody_write_pgm(os.fsencode(filename), img.ctypes.data, img.shape[1], img.shape[0]))


def partition(n, rank, world):
    """(lo, cnt) of rank's own range when n bodies are partitioned over world ranks (nbody_partition)."""
    lo, cnt = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib.nbody_partition(n, rank, world, ctypes.byref(lo), ctypes.byref(cnt)))
    return lo.value, cnt.value


def comm_unique_id():
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib.nbody_comm_unique_id(buf))
    return buf.raw


class Stepper:
    """Device-resident stepper: the loop body of src/nbody.cu:460-545 without the per-step host round trip."""

    def __init__(self, cfg=None, capacity=None, precision=F32, semantics=LITERAL, device=0, rank=0, world=1,
                 record_events=False, group=False, comm_id=None, timestep=None, growthRate=None,
                 fieldWidth=None, fieldHeight=None, event_capacity=0, kernel_variant=0, force_comm=False):
        d = _CtxDesc()
        if cfg is not None:
            lib.nbody_ctx_desc_from_config(ctypes.byref(d), ctypes.byref(cfg), precision)
        d.precision, d.semantics, d.device, d.rank, d.world = precision, semantics, device, rank, world
        if capacity is not None:
            d.capacity = capacity
        for name, val in (("timestep", timestep), ("growthRate", growthRate), ("fieldWidth", fieldWidth),
                          ("fieldHeight", fieldHeight)):
            if val is not None:
                setattr(d, name, val)
        d.flags = ((FLAG_RECORD_EVENTS if record_events else 0) | (FLAG_GROUP_EXCHANGE if group else 0) |
                   (FLAG_FORCE_COMM if force_comm else 0))
        d.event_capacity = event_capacity
        d.kernel_variant = kernel_variant
        self._comm_id = ctypes.create_string_buffer(comm_id, COMM_ID_BYTES) if comm_id else None
        d.comm_id = ctypes.cast(self._comm_id, ctypes.c_void_p) if self._comm_id else None
        self.precision = precision
        self.capacity = d.capacity
        self.world, self.rank = world, rank
        self._ctx = ctypes.c_void_p()
        _check(lib.nbody_ctx_create(ctypes.byref(self._ctx), ctypes.byref(d)))

    def close(self):
        if getattr(self, "_ctx", None) and lib is not None:     # `lib` is gone at interpreter shutdown
            lib.nbody_ctx_destroy(self._ctx)
            self._ctx = None

    def render_image(self, width, height):
        """cudaMemset(254) + generateImage + D2H, src/nbody.cu:531-537."""
        img = np.zeros((height, width), dtype=np.uint8)
        _check(lib.nbody_render_image(self._ctx, img.ctypes.data, width, height))
        return img

    def save_state(self, path):
        _check(lib.nbody_state_save(self._ctx, os.fsencode(path)))

    def load_state(self, path):
        _check(lib.nbody_state_load(self._ctx, os.fsencode(path)))

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, bodies):
        """BodiesData::uploadToDevice, src/nbody.cu:88-96."""
        assert bodies.precision == self.precision
        _check(lib.nbody_upload(self._ctx, bodies.ptr, bodies.numBodies))

    def step(self, nsteps=1):
        _check(lib.nbody_step(self._ctx, nsteps))

    def sync(self):
        _check(lib.nbody_sync(self._ctx))

    def download(self):
        out = BodiesData(0, self.precision, self.capacity)
        n = ctypes.c_int(0)
        _check(lib.nbody_download(self._ctx, out.ptr, ctypes.byref(n)))
        out.numBodies = n.value
        return out

    def body_count(self):
        n = ctypes.c_int(0)
        _check(lib.nbody_body_count(self._ctx, ctypes.byref(n)))
        return n.value

    def own_range(self):
        lo, cnt = ctypes.c_int(0), ctypes.c_int(0)
        _check(lib.nbody_own_range(self._ctx, ctypes.byref(lo), ctypes.byref(cnt)))
        return lo.value, cnt.value

    def events(self, cap=1 << 20):
        buf = np.zeros(cap, dtype=EVENT_DTYPE)
        total = ctypes.c_int64(0)
        _check(lib.nbody_get_events(self._ctx, buf.ctypes.data, cap, ctypes.byref(total)))
        if total.value > cap:
            raise NbodyError(-7, "event log holds %d events, buffer %d" % (total.value, cap))
        return buf[:total.value]

    def clear_events(self):
        _check(lib.nbody_clear_events(self._ctx))

    def force_kernel_name(self):
        return lib.nbody_force_kernel_name(self._ctx).decode()

    def set_kernel_timing(self, enable=True):
        _check(lib.nbody_set_kernel_timing(self._ctx, int(enable)))

    def force_only(self, reps):
        _check(lib.nbody_debug_force_only(self._ctx, reps))

    def ring_probe(self):
        out = (ctypes.c_uint64 * 8)()
        _check(lib.nbody_debug_ring_probe(self._ctx, ctypes.byref(out)))
        return list(out)

    def stats(self):
        s = Stats()
        _check(lib.nbody_get_stats(self._ctx, ctypes.byref(s)))
        return s


class StepperGroup:
    """All ranks of a partition as contexts of this process (nbody_group_step): one per device, or several
    on one device.  The exchange is stream-ordered peer copies, no RCCL."""

    def __init__(self, world, devices=None, **kw):
        devices = devices or [0] * world
        self.world = world
        self.ranks = [Stepper(rank=g, world=world, device=devices[g], group=world > 1, **kw)
                      for g in range(world)]
        self._arr = (ctypes.c_void_p * world)(*[r._ctx for r in self.ranks])
        self.precision = self.ranks[0].precision
        self.capacity = self.ranks[0].capacity

    def upload(self, bodies):
        for r in self.ranks:
            r.upload(bodies)

    def step(self, nsteps=1):
        _check(lib.nbody_group_step(self._arr, self.world, nsteps))

    def download(self):
        out = BodiesData(0, self.precision, self.capacity)
        n = ctypes.c_int(0)
        _check(lib.nbody_group_download(self._arr, self.world, out.ptr, ctypes.byref(n)))
        out.numBodies = n.value
        return out

    def close(self):
        for r in self.ranks:
            r.close()
