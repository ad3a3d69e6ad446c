#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profile_round.sh (gpurun_out/<R>_*) into the summaries kept under profiles/:
   <R>_bench_kernel_stats.csv, <R>_bench_under_rocprof.json, <R>_traffic_pmc.json, <R>_pmc_ring.txt, <R>_pmc_f64.txt.
   python3 profiles_from_round.py r03"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counters(d, want="forces"):
    f = glob.glob("%s/%s/*counter_collection.csv" % (G, d)) + glob.glob("%s/%s/*/*counter_collection.csv" % (G, d))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if want not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    return acc, cnt


def per_kernel_mean(d, counter):
    f = glob.glob("%s/%s/*counter_collection.csv" % (G, d)) + glob.glob("%s/%s/*/*counter_collection.csv" % (G, d))
    by = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> dispatch -> value (summed over XCDs)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            by[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {"launches": len(v), "mean_KiB": sum(v.values()) / len(v)} for k, v in by.items()}


def last_line(name):
    for line in reversed(open(os.path.join(G, name)).read().splitlines()):
        if line.startswith("N="):
            return line
    return ""


# (a) kernel stats + the bench line taken under the profiler
shutil.copy(os.path.join(G, "%s_prof_bench/run_kernel_stats.csv" % R), os.path.join(P, "%s_bench_kernel_stats.csv" % R))
for line in open(os.path.join(G, "%s_prof_bench.json" % R)):
    if line.startswith("{"):
        open(os.path.join(P, "%s_bench_under_rocprof.json" % R), "w").write(line)

# (b) traffic
fetch, write = per_kernel_mean("%s_pmc_fetch" % R, "FETCH_SIZE"), per_kernel_mean("%s_pmc_write" % R, "WRITE_SIZE")
ring = [k for k in fetch if "forces_ring" in k][0]
fk, wk = fetch[ring]["mean_KiB"] * 1024 * 2, write[ring]["mean_KiB"] * 1024
n = 262144
cal = {k: fetch[k]["mean_KiB"] for k in fetch if any(s in k for s in ("records_to_tiles", "compact_count", "unpack_slots"))}
out = {
    "round": int(R[1:]), "workload": "bench.py default: N=262144 fp32 radii 0, 1 GPU, kernel %s" % ring,
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline (and a second, separate pass with --pmc WRITE_SIZE): csrc/tune/profile_round.sh; summarised "
               "by csrc/tune/profiles_from_round.py",
    "raw": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
    "correction": "MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB at the L2s' fabric side; on gfx950 "
                  "FETCH_SIZE reports half the bytes of line-granular reads.  Calibrated in THIS pass on kernels with known "
                  "bytes: records_to_tiles_f32 reads the 4 MiB replica, compact_count one word of every line of the 4 MiB "
                  "staged records, unpack_slots 24 bytes per body (6 MiB): reported %s KiB - half each - so FETCH is doubled "
                  "for the force kernel too; WRITE_SIZE is exact.  Infinity-Cache hits are included: an upper bound on HBM bytes."
                  % json.dumps({k: round(v) for k, v in cal.items()}),
    "forces_kernel_traffic_bytes_per_launch": fk + wk, "forces_kernel_fetch_bytes_corrected": fk,
    "forces_kernel_write_bytes": wk, "algorithmic_bytes_per_launch": 48 * n,
    "reading": "%.1f MB per launch = %.1fx the %.1f MB compulsory bytes, at the floor of this fabric-side counter: 8 XCD-private "
               "L2s each pull the x, y, m planes of the replica once per launch (8 x 3.15 MB) plus the own bodies' velocities; "
               "the Infinity Cache serves it (the replica is rewritten once per step and then only read)."
               % ((fk + wk) / 1e6, (fk + wk) / (48 * n), 48 * n / 1e6)}
json.dump(out, open(os.path.join(P, "%s_traffic_pmc.json" % R), "w"), indent=1)

# (c) SQ counters of the ring kernel
lines = ["# rocprofv3 --pmc, three passes per shape (csrc/tune/profile_round.sh (c)), one rank's force kernel launched back to back",
         "# (rank_kernel.py): g1 = N=262144 on one GPU (4 rings x 4 waves), g8 = rank 3 of 8 at N=262144 (2 x 8), n64k = N=65536 (4 x 4)"]
shapes = {"g1": (262144, 1), "g8": (262144, 8), "n64k": (65536, 1)}
reading = []
for tag in ("g1", "g8", "n64k"):
    tot = {}
    for x in ("x1", "x2", "x3"):
        acc, cnt = counters("%s_pmc_ring_%s_%s" % (R, tag, x))
        for k in acc:
            lines.append("%s_pmc_ring_%s_%s %s" % (R, tag, x, k[:70]))
            for c, v in sorted(acc[k].items()):
                lines.append("   %-28s %.4g (n=%d)" % (c, v, cnt[(k, c)]))
                tot[c] = (v, cnt[(k, c)])
    ln = last_line("%s_pmc_ring_%s_x3.log" % (R, tag))
    lines.append(ln)
    pairs = float(ln.split("launch, ")[1].split(" pairs")[0])
    ms = float(ln.split(": ")[1].split(" ms")[0])
    valu, nl = tot["SQ_INSTS_VALU"]
    act, na = tot["SQ_ACTIVE_INST_VALU"]
    gui, ng = tot["GRBM_GUI_ACTIVE"]
    conf = tot["SQ_LDS_BANK_CONFLICT"][0]
    reading.append("#   %-5s %.2f VALU instructions per pair (SQ_INSTS_VALU / (pairs / 64)); VALU pipe occupancy SQ_ACTIVE_INST_VALU*4 / "
                   "(1024 SIMDs * GRBM_GUI_ACTIVE/8) = %.3f; LDS bank conflicts %.3g; clock GRBM_GUI_ACTIVE/8/t = %.2f GHz (kernel %.3f ms under the profiler)"
                   % (tag, valu / nl / (pairs / 64), (act / na) * 4 / (1024 * gui / ng / 8), conf, gui / ng / 8 / (ms * 1e-3) / 1e9, ms))
lines.append("# Reading (per launch):")
lines += reading
open(os.path.join(P, "%s_pmc_ring.txt" % R), "w").write("\n".join(lines) + "\n")

# (d) fp64
lines = ["# rocprofv3 --pmc, two passes per shape (profile_round.sh (d)): the fp64 production kernel forces_v3w_f64, one rank launched back to back:",
         "#   c5g1 = N=1048576 on one GPU, c5g8 = rank 4 of 8 at N=1048576 (131072 own bodies: C5's 8-rank shape, priority rotation on)"]
reading = []
for tag in ("c5g1", "c5g8"):
    tot = {}
    for x in ("x1", "x2"):
        acc, cnt = counters("%s_pmc_f64_%s_%s" % (R, tag, x))
        for k in acc:
            lines.append("%s_pmc_f64_%s_%s %s" % (R, tag, x, k[:70]))
            for c, v in sorted(acc[k].items()):
                lines.append("   %-28s %.4g (n=%d)" % (c, v, cnt[(k, c)]))
                tot[c] = (v, cnt[(k, c)])
    ln = last_line("%s_pmc_f64_%s_x2.log" % (R, tag))
    lines.append(ln)
    pairs = float(ln.split("launch, ")[1].split(" pairs")[0])
    ms = float(ln.split(": ")[1].split(" ms")[0])
    valu, nl = tot["SQ_INSTS_VALU"]
    act, na = tot["SQ_ACTIVE_INST_VALU"]
    gui, ng = tot["GRBM_GUI_ACTIVE"]
    conf, nc = tot["SQ_LDS_BANK_CONFLICT"]
    idx, ni = tot["SQ_LDS_IDX_ACTIVE"]
    reading.append("#   %-5s %.2f VALU instructions per pair; VALU pipe occupancy %.3f; LDS: %.3g conflict cycles of %.3g active (%.0f %%); "
                   "20 flop x pairs / t = %.1f TFLOP/s = %.3f of the 78.6 TFLOP/s fp64 vector peak (kernel %.1f ms under the profiler)"
                   % (tag, valu / nl / (pairs / 64), (act / na) * 4 / (1024 * gui / ng / 8), conf / nc, idx / ni, 100 * conf / idx,
                      20 * pairs / (ms * 1e-3) / 1e12, 20 * pairs / (ms * 1e-3) / 1e12 / 78.6, ms))
lines.append("# Reading (per launch):")
lines += reading
open(os.path.join(P, "%s_pmc_f64.txt" % R), "w").write("\n".join(lines) + "\n")
print("written:", [f for f in sorted(os.listdir(P)) if f.startswith(R)])
