#!/usr/bin/env python3
"""bench.py -- body-pair-interactions/sec of the HIP stepper (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (forces + collisions + drift + compaction [+ slot all-gather]) over the
whole body set.  Workload = BASELINE.json configs[3] shape: N=262144 bodies, fp32, the reference's own initial
condition (seed 1024), radii 0 (the "positions-only exchange" headline row of SURVEY.md 8d), literal
reference semantics.  N>1: STRONG scaling - the same 262144 bodies range-partitioned over the ranks, one
process per GPU, the per-step all-gather over RCCL inside the library; torch.distributed (gloo) only carries
the 128-byte communicator id, the barriers and the max-over-ranks time.  After the timed region a multi-rank run
checks ITSELF: the state the ranks hold (collective download) against a single-rank run of the same steps, bit for
bit (`parity`), so a scaling line always carries its own correctness bit.

Next to the CPU baseline (the oracle on the host cores, `cpu_baseline`) a 1-GPU fp32 run reports the speed of the
reference's OWN kernels on the same GPU (`reference_kernels_on_this_gpu`: src/nbody.cu's device code compiled unmodified
by hipcc, oracle/ref_hip) when that checker library has been built.

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's multi-process set-up fails with
# `hipIpcGetMemHandle: invalid argument` (must be in the environment before anything initialises HIP)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FLOP_PER_PAIR = 20.0            # SURVEY.md 8d convention (18 + sqrt + divide)
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (packed issue)
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s


TRAFFIC_FILE = os.path.join("profiles", "r03_traffic_pmc.json")


def measured_traffic(a, world):
    """HBM-side bytes per force-kernel launch: NOT measured in this run (PMC counters need rocprofv3) but read from
    the committed rocprofv3 passes of this same command (profiles/, collected in separate --pmc passes and corrected
    as MI355X_MICROARCH.md prescribes).  Only reported for the configuration it was measured on; the JSON line
    names the file it came from."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if (world != 1 or a.bodies != 262144 or a.fp64 or a.stock_radii or a.variant != 0 or a.clean or
            not os.path.exists(path)):
        return None, None
    return json.load(open(path))["forces_kernel_traffic_bytes_per_launch"], TRAFFIC_FILE


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology():
    """Sockets, physical cores and logical CPUs of the host (/proc/cpuinfo), and the CPUs this process may run on:
    BASELINE.md asks for the core count behind the CPU figure - OpenMP threads alone do not say whether they are
    cores or SMT siblings."""
    sockets, cores, logical = set(), set(), 0
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                logical += 1
                phys = core = None
            elif key == "physical id":
                phys = val
                sockets.add(val)
            elif key == "core id":
                core = val
            if phys is not None and core is not None:
                cores.add((phys, core))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    # the CPU-time quota of the container (cgroup v2 cpu.max / v1 cfs quota), in cores: what the OpenMP threads SHARE
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    return {"sockets": len(sockets) or None, "physical_cores": len(cores) or None, "logical_cpus": logical or None,
            "cpus_usable_by_this_process": usable, "cpu_quota_cores": quota}


def cpu_baseline(nb, bodies, cfg, budget_s=12.0, semantics=0):
    """Times the CPU oracle (oracle/nbody_oracle.c, OpenMP over i) on a bounded sample of the same workload:
    a contiguous range of i-bodies of step 1, every one against all its j's.  Checker used as the reported
    baseline only (kind 'port': the reference has no CPU stepper)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    n = bodies.numBodies
    topo = cpu_topology()
    threads = ol.port().oracle_get_max_threads()
    if topo["cpu_quota_cores"] and topo["cpu_quota_cores"] < threads:
        # more threads than the container's CPU quota only time-share the same cores: use (and report) the quota
        threads = max(1, int(round(topo["cpu_quota_cores"])))
        ol.port().oracle_set_threads(threads)
    blk = bodies.contiguousData
    dt, gr = np.float32(cfg.timestep), np.float32(cfg.growthRate)
    if bodies.precision == nb.F64:
        dt, gr = float(dt), float(gr)
    probe = min(n, 32 * threads)
    t0 = time.perf_counter()
    *_, st = ol.port_range(blk, n, 0, probe, dt, cfg.fieldWidth, cfg.fieldHeight, gr, semantics=semantics)
    t_probe = time.perf_counter() - t0
    rate = st.pairs / max(t_probe, 1e-9)
    per_body = st.pairs / probe
    count = int(max(threads * 4, min(n, budget_s * rate / per_body)))
    count = min(n, count - count % threads if count > threads else count)
    lo = (n // 2 // 128) * 128
    lo = min(lo, n - count)
    t0 = time.perf_counter()
    oP, oV, oM, oR, _, st = ol.port_range(blk, n, lo, lo + count, dt, cfg.fieldWidth, cfg.fieldHeight, gr,
                                          semantics=semantics)
    t = time.perf_counter() - t0
    base = {"value": st.pairs / t, "unit": "body-pair-interactions/sec", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "cpu_topology": topo,
            "cores_note": "`cores` = OpenMP threads used; the host has %s socket(s), %s physical cores, %s logical CPUs, "
                          "%s usable by this process, CPU quota of the container: %s cores" %
                          (topo["sockets"], topo["physical_cores"], topo["logical_cpus"],
                           topo["cpus_usable_by_this_process"], topo["cpu_quota_cores"]),
            "sample": "bodies [%d,%d) of step 1 at N=%d against all j (%d pairs, %.1f s, OpenMP %d threads)" %
                      (lo, lo + count, n, st.pairs, t, threads)}
    # the same restatement on ONE thread, about two seconds of it
    ol.port().oracle_set_threads(1)
    one = max(8, min(count, int(10.0 * rate / threads / per_body)))   # ~2 s: one thread is faster than rate / threads
    t0 = time.perf_counter()
    *_, st1 = ol.port_range(blk, n, lo, lo + one, dt, cfg.fieldWidth, cfg.fieldHeight, gr, semantics=semantics)
    t1 = time.perf_counter() - t0
    ol.port().oracle_set_threads(threads)
    base["single_thread"] = {"value": st1.pairs / t1, "unit": "body-pair-interactions/sec",
                             "sample": "bodies [%d,%d) of the same step (%d pairs, %.1f s)" % (lo, lo + one, st1.pairs, t1)}
    return base, parity_of_sample(nb, bodies, cfg, lo, count, oP, oV, oM, oR, semantics)


def parity_of_ranks(nb, st, bodies, cfg, precision, device, total_steps, rank, semantics=0, dist=None):
    """world > 1: the state the ranks hold after the run (collective download over RCCL: every rank calls it) against
    a fresh SINGLE-rank stepper run for the same number of steps on rank 0's GPU.  Sharding changes who computes a
    body, never what is computed, so the bar is bitwise equality; the single-rank path is the one the 1-GPU line checks
    against the CPU oracle."""
    import numpy as np
    if dist is not None:
        # The download is a collective: a rank whose context has already failed must not leave the others waiting in
        # it.  Every rank looks at its own context first and all agree (gloo) on whether to go in.
        import torch
        try:
            st.sync()
            ok = 1
        except nb.NbodyError:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            raise RuntimeError("a rank's context reports a device failure: collective download skipped on every rank")
    got = st.download()
    if rank != 0:
        return None
    one = nb.Stepper(cfg, precision=precision, device=device, semantics=semantics)
    one.upload(bodies)
    one.step(total_steps)
    want = one.download()
    one.close()
    u = np.uint64 if precision == nb.F64 else np.uint32
    same_n = got.numBodies == want.numBodies
    equal = bool(same_n and np.array_equal(got.block.view(u), want.block.view(u)))
    out = {"against": "single-rank run of the same %d steps on rank 0's GPU (itself checked against the CPU oracle by "
                      "the 1-GPU line and the test-suite)" % total_steps,
           "bodies_after": int(got.numBodies), "bodies_after_single_rank": int(want.numBodies),
           "bitwise_equal": equal}
    if same_n:
        out["max_abs_dpos"] = float(np.abs(got.Positions.astype(np.float64) - want.Positions).max())
        out["max_abs_dvel"] = float(np.abs(got.Velocities.astype(np.float64) - want.Velocities).max())
    return out


def reference_kernels_on_gpu(nb, bodies, cfg, our_ms_per_step, steps=2):
    """The reference's OWN kernels on this GPU, as a baseline next to the CPU one: oracle/_ref/libnbody_ref_hip.so is the
    reference's device code (src/nbody.cu:126-292) compiled unmodified by hipcc for gfx950 (oracle/ref_hip, built where
    /root/reference exists) and launched with the reference's geometry.  Timed: ComputeForces + MoveBodies with HIP
    events, i.e. WITHOUT the reference loop's per-step cudaMalloc / PCIe round trip / host compaction.  fp64 runs use the
    fp64 reading of the same text (the reference has no fp64 kernel).  Checker / baseline only, after the timed region;
    absent library -> None."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    f64 = bodies.precision == nb.F64
    if not (ol.have_ref_hip_f64() if f64 else ol.have_ref_hip()):
        return None
    n = bodies.numBodies
    blk = bodies.contiguousData.copy()
    pairs = 0
    cur = n
    total_ms = 0.0
    dt, gr = np.float32(cfg.timestep), np.float32(cfg.growthRate)
    if f64:
        dt, gr = float(dt), float(gr)
    for _ in range(steps):
        pairs += ol.port().oracle_pairs_per_step(cur, ol.LITERAL)
        cur, ms, _ = ol.ref_hip_run(blk, cur, 1, dt, cfg.fieldWidth, cfg.fieldHeight, gr)
        total_ms += ms
    return {"what": "the reference's ComputeForces + MoveBodies (src/nbody.cu:139-292)%s, compiled by hipcc "
                    "-O3 for gfx950, its own launch geometry, kernel time only, first %d steps of this workload" %
                    (" with `float` read as `double` (the reference has no fp64 code: oracle/ref_hip, REF_FLOAT_AS_DOUBLE)"
                     if f64 else ", unmodified", steps),
            "ms_per_step": total_ms / steps, "value": pairs / (total_ms * 1e-3), "unit": "body-pair-interactions/sec",
            "this_framework_ms_per_step": our_ms_per_step, "speedup": (total_ms / steps) / our_ms_per_step}


def parity_of_sample(nb, bodies, cfg, lo, count, oP, oV, oM, oR, semantics=0):
    """The second half of BASELINE.json's metric ("max-|dpos| vs ref"): the HIP path's state after step 1
    against the oracle's for the bodies the CPU leg just computed (outside the timed region).  The device
    compacts deleted bodies away, so post-step indices are mapped through the device's own deletion log."""
    import numpy as np
    st = nb.Stepper(cfg, precision=bodies.precision, record_events=True, event_capacity=1 << 23, semantics=semantics)
    st.upload(bodies)
    st.step(1)
    out = st.download()
    ev = st.events()
    st.close()
    deleted = np.unique(ev["i"][ev["kind"] == 1]).astype(np.int64)
    keep = oM != 0
    pre = lo + np.nonzero(keep)[0]
    post = pre - np.searchsorted(deleted, pre)
    same_deleted = np.array_equal(deleted[(deleted >= lo) & (deleted < lo + count)], lo + np.nonzero(~keep)[0])
    gP, gV, gM, gR = out.Positions[post], out.Velocities[post], out.Masses[post], out.Radii[post]
    u = np.uint64 if oP.dtype == np.float64 else np.uint32
    bitwise = bool(same_deleted and all(np.array_equal(np.ascontiguousarray(g).view(u), np.ascontiguousarray(w).view(u))
                                        for g, w in ((gP, oP[keep]), (gV, oV[keep]), (gM, oM[keep]), (gR, oR[keep]))))
    return {"against": "CPU oracle, step 1, bodies [%d,%d)" % (lo, lo + count), "bodies": int(count),
            "deleted_in_sample": int((~keep).sum()), "deleted_sets_equal": bool(same_deleted),
            "max_abs_dpos": float(np.abs(gP.astype(np.float64) - oP[keep]).max()),
            "max_abs_dvel": float(np.abs(gV.astype(np.float64) - oV[keep]).max()),
            "bitwise_equal": bitwise}


def baseline_config_name(a):
    """Which entry of BASELINE.json's `configs` the workload has the shape of (the metric is quoted on [3])."""
    if a.fp64:
        return "BASELINE.json configs[4] shape: " if a.bodies == 1048576 and not a.stock_radii else ""
    if a.bodies == 262144 and not a.stock_radii:
        return "BASELINE.json configs[3] (the metric's configuration): "
    if a.bodies == 65536:
        return "BASELINE.json configs[2] shape: " if a.stock_radii else "BASELINE.json configs[1] shape: "
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--stock-radii", action="store_true", help="radii 50-200 (collisions on) instead of 0")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--clean", action="store_true",
                    help="clean semantics (every body active, true all-pairs, j ascending: SURVEY.md 8 f4) instead of "
                         "the reference's literal index semantics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="world > 1: skip the comparison with a single-rank run")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--force-comm", action="store_true",
                    help="take the multi-rank code path (rendezvous, RCCL communicator, slot all-gather) even "
                         "with one rank; used to rehearse the N>1 launch on a 1-GPU box")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if a.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: start one rank per GPU as child processes (nothing in this
        # process has touched the GPU yet) and leave with the launcher's exit code
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # Exactly ONE line goes to stdout: libraries write there too (RCCL prints a version banner when its first
    # communicator comes up), so from here on file descriptor 1 is stderr and the JSON line goes to the saved one.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch                       # plumbing: rendezvous, barriers, device selection
    import ppa_nbody_collisions_amd as nb
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible")
    if world > 1 and ndev > 1 and world > ndev:
        raise SystemExit("bench.py: %d ranks but only %d HIP devices visible (RCCL needs one device per rank)" % (world, ndev))
    local_rank %= ndev                 # a launcher that exposes one device per rank (HIP_VISIBLE_DEVICES)

    comm_id = None
    dist = None
    if world > 1 or (a.force_comm and "RANK" in os.environ):
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        torch.cuda.set_device(local_rank)
        box = [nb.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm_id = box[0]
    else:
        torch.cuda.set_device(local_rank)

    precision = nb.F64 if a.fp64 else nb.F32
    kw = {} if a.stock_radii else {"minRadius": 0.0, "maxRadius": 0.0}
    cfg = nb.stock_config(particleCount=a.bodies, totalIterations=a.steps, **kw)
    bodies = nb.init_bodies(cfg, precision)
    if a.force_comm and comm_id is None:
        comm_id = nb.comm_unique_id()
    semantics = nb.CLEAN if a.clean else nb.LITERAL
    st = nb.Stepper(cfg, precision=precision, device=local_rank, rank=rank, world=world, comm_id=comm_id,
                    kernel_variant=a.variant, force_comm=a.force_comm, semantics=semantics)
    st.upload(bodies)

    def barrier():
        torch.cuda.synchronize()
        st.sync()
        if dist is not None:
            dist.barrier()

    st.step(a.warmup)
    barrier()
    st.set_kernel_timing(True)          # creates the timing events HERE: the timed region only records them
    s0 = st.stats()
    barrier()
    t0 = time.perf_counter()
    st.step(a.steps)
    st.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    s1 = st.stats()
    st.set_kernel_timing(False)

    kernel_name = st.force_kernel_name()
    pairs = s1.pairs - s0.pairs
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        p = torch.tensor([pairs], dtype=torch.int64)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        pairs = int(p[0])

    if rank == 0:
        value = pairs / elapsed
        launches = max(1, s1.force_kernel_launches)
        k_ms = s1.force_kernel_ms / launches                  # HIP events on the launch stream
        k_pairs = (s1.pairs - s0.pairs) / launches            # pairs one launch of THIS rank evaluates
        real = 8 if a.fp64 else 4
        # ALGORITHMIC (compulsory) bytes of one force-kernel launch, SURVEY.md 8d / DESIGN.md 4.1: read the {x,y,m,r}
        # replica once (4 reals per body of the whole set) + read V (2) and write the staged record (4) and V (2) of
        # the own range.  One rank: (4 + 8) reals * N = 48*N bytes in fp32, 96*N in fp64.
        alg_bytes = 4.0 * real * s1.n_bodies + 8.0 * real * (s1.n_own if world > 1 else s1.n_bodies)
        peak_valu = PEAK_FP64_VALU_TFLOPS if a.fp64 else PEAK_FP32_VALU_TFLOPS
        traffic, traffic_source = measured_traffic(a, world)
        out = {
            "metric": "body-pair-interactions/sec at N=%d" % a.bodies,
            "value": value, "unit": "body-pair-interactions/sec", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if a.fp64 else "f32",
            "data": "synthetic",
            "config": {"workload": "%sN=%d bodies, %s, %s, reference initial condition (seed 1024), %s step semantics, "
                                   "%d steps" %
                                   ("" if a.clean else baseline_config_name(a), a.bodies, "fp64" if a.fp64 else "fp32",
                                    "stock radii 50-200 (collisions on)" if a.stock_radii else "radii 0",
                                    "CLEAN (every body active, true all-pairs: SURVEY.md 8 f4; not the reference's)"
                                    if a.clean else "literal reference", a.steps),
                       "bodies_after": s1.n_bodies,
                       "parallelism": "range-partition x%d, RCCL slot all-gather per step" % world
                       if world > 1 else "single GPU"},
            # north_star asks for the HBM fraction; the roof that BINDS this kernel is the fp32/fp64 vector ALU
            # (arithmetic intensity ~1e5 flop/B): its fraction is roofline.valu, computed from the same live kernel time
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms": k_ms, "kernel_launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes, "binding": "valu",
                         "valu": {"bound": "valu_fp32" if not a.fp64 else "valu_fp64",
                                  "achieved": FLOP_PER_PAIR * k_pairs / (k_ms * 1e-3) / 1e12, "peak": peak_valu,
                                  "unit": "TFLOP/s",
                                  "frac": FLOP_PER_PAIR * k_pairs / (k_ms * 1e-3) / 1e12 / peak_valu,
                                  "flop_per_pair": FLOP_PER_PAIR, "pairs_per_launch": k_pairs}},
        }
        if world > 1 or a.force_comm:
            # where a multi-rank step goes: the force kernel and the slot all-gather are timed with HIP events on the
            # context's stream (rank 0's), the rest is compaction / unpack / launch gaps / waiting for slower ranks
            n_x = max(1, s1.exchange_launches)
            x_ms = s1.exchange_ms / n_x
            out["step_breakdown"] = {
                "rank": 0, "force_kernel_ms": k_ms, "exchange_ms": x_ms,
                "rest_ms": out["ms_per_step"] - k_ms - x_ms,
                "exchange_bytes_received_per_step": (s1.exchange_bytes - s0.exchange_bytes) / max(1, a.steps),
                "slot_bytes_per_rank_now": s1.slot_bytes_now,
                "note": "exchange = one all-gather of {count | records | velocities} slots laid out for the live bound of "
                        "the body count (24 B per body + 32 B header per rank)"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(nb, bodies, cfg, a.cpu_budget, semantics)
            try:
                ref_gpu = None if a.clean else reference_kernels_on_gpu(nb, bodies, cfg, out["ms_per_step"])
            except Exception as e:        # a baseline leg must never take the benchmark line down with it
                ref_gpu = {"error": "%s: %s" % (type(e).__name__, e)}
            if ref_gpu is not None:
                out["reference_kernels_on_this_gpu"] = ref_gpu
    if (world > 1 or a.force_comm) and not a.no_parity:
        # every rank takes part (the download is a collective); rank 0 compares and reports.  A failing check must
        # not take the measured line down with it: it is reported in the line instead.
        try:
            par = parity_of_ranks(nb, st, bodies, cfg, precision, local_rank, a.warmup + a.steps, rank, semantics, dist)
        except Exception as e:
            par = {"error": "%s: %s" % (type(e).__name__, e), "bitwise_equal": False}
        if rank == 0:
            out["parity" if world > 1 else "parity_rccl_path"] = par
            out["config"]["rccl_ranks"] = world
    if rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    st.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
