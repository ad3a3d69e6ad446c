#!/usr/bin/env python3
"""Development helper: print the hottest basic block (most v_rsq_f32) of a kernel in build/csrc/nbody_ctx.s
with an opcode histogram.   usage: dump_loop.py <kernel-name-substring> [--full]"""
import collections
import re
import sys

s = open("/root/repo/build/csrc/nbody_ctx.s").read()
sub = sys.argv[1]
lines = s.split("\n")
starts = [k for k, l in enumerate(lines) if l.startswith("_ZN") and sub in l and ": ;" in l]
k0 = starts[0]
k1 = next(k for k in range(k0, len(lines)) if lines[k].startswith(".Lfunc_end"))
body = "\n".join(lines[k0:k1])
blocks = re.split(r"\n(?=\.LBB)", body)
best = max(blocks, key=lambda b: b.count("v_rsq_f32"))
ops = [l.split()[0] for l in best.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
print(len(ops), "instructions;", best.count("v_rsq_f32"), "pairs per trip")
print(collections.Counter(ops).most_common(50))
if "--full" in sys.argv:
    print(best)
