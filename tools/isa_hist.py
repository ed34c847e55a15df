#!/usr/bin/env python3
"""Instruction histogram of a line range of hipcc's gfx950 assembly (hipcc -S --cuda-device-only): how the VALU budget of a hot loop divides
into multiply-adds, the per-product bookkeeping and everything else.    python tools/isa_hist.py file.s first_line last_line [first last ...]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
tot = collections.Counter()
for a, b in zip(sys.argv[2::2], sys.argv[3::2]):
    for ln in lines[int(a) - 1:int(b)]:
        m = re.match(r"\s+([a-z][a-z0-9_]+)\s", ln + " ")
        if m and not ln.strip().startswith((";", ".")):
            tot[m.group(1)] += 1
valu = sum(v for k, v in tot.items() if k.startswith("v_"))
print("total %d, VALU %d, SALU %d, memory %d" % (sum(tot.values()), valu, sum(v for k, v in tot.items() if k.startswith("s_")),
                                                  sum(v for k, v in tot.items() if k.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")))))
for k, v in tot.most_common(40):
    print("%6d  %s" % (v, k))
