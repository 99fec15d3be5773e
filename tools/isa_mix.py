#!/usr/bin/env python3
"""Static instruction mix of one kernel of a translation unit (hipcc -S, gfx950): how many vector-ALU instructions are
arithmetic, selects / compares (the in-bounds predicates of the bilinear taps), conversions, cross-lane moves; how many
scalar and memory instructions.  Used to price re-formulations of k_geom_point_fwd before building them
(profiles/r03_point_fwd_isa_mix.md).

    python tools/isa_mix.py unsupervised_depth_opticalflow_egomotion_amd/csrc/loss_stack_fwd.hip k_geom_point_fwdILb0E
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only".split()

CLASSES = [
    ("fp32 with DPP (cross-lane sums)", r"^v_\w+_dpp$"),
    ("fp32 arithmetic (mul / add / sub / fma / fmac / mad)", r"^v_(mul|add|sub|subrev|fma|fmac|mad|mac|fmamk|fmaak)_f32"),
    ("select (v_cndmask)", r"^v_cndmask"),
    ("compare (v_cmp*)", r"^v_cmp"),
    ("min / max / med3", r"^v_(min|max|med3)"),
    ("transcendental (rcp / rsq / sqrt / exp / log)", r"^v_(rcp|rsq|sqrt|exp|log)"),
    ("conversion / floor / ldexp / frexp", r"^v_(cvt|floor|ceil|trunc|rndne|ldexp|frexp|fract)"),
    ("integer / address arithmetic", r"^v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|bfe|bfi|not|lshlrev|lshrrev|ashrrev|add_lshl|lshl_add|mul_lo|mul_hi|mad_u|mad_i|add3|lshl_or|and_or|or3)"),
    ("moves (v_mov / readlane / readfirstlane / writelane)", r"^v_(mov|readlane|readfirstlane|writelane|swap|perm|accvgpr)"),
]


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN\S*%s\S*:" % re.escape(pat), l))
    end = next(i for i in range(start, len(text)) if text[i].startswith("\t.amdhsa_kernel") or text[i].startswith(".Lfunc_end"))
    ops = collections.Counter()
    for l in text[start:end]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m:
            ops[m.group(1)] += 1
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    print("kernel label: %s" % text[start].split(":")[0])
    print("\n| class | static count | share of vector ALU |\n|---|---|---|")
    left = dict(valu)
    total = sum(valu.values())
    for name, rx in CLASSES:
        n = 0
        for k in list(left):
            if re.match(rx, k):
                n += left.pop(k)
        print("| %s | %d | %.1f %% |" % (name, n, 100.0 * n / total))
    rest = sum(left.values())
    print("| other vector ALU (%s) | %d | %.1f %% |" % (", ".join(sorted(left)[:6]), rest, 100.0 * rest / total))
    print("| **vector ALU total** | **%d** | |" % total)
    for name, rx in (("scalar ALU / control (s_*)", r"^s_(?!waitcnt|nop|load|buffer_load)"), ("s_waitcnt", r"^s_waitcnt"), ("s_nop", r"^s_nop"),
                     ("scalar loads", r"^s_(load|buffer_load)"), ("global loads", r"^global_load"), ("global stores", r"^global_store"),
                     ("LDS (ds_*)", r"^ds_")):
        print("| %s | %d | |" % (name, sum(v for k, v in ops.items() if re.match(rx, k))))


if __name__ == "__main__":
    main()
