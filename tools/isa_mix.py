#!/usr/bin/env python3
"""Instruction mix of a kernel from the compiler's assembly, priced with the issue
rates tools/micro/valu_rate.hip measures on the chip:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \\
        -I include xvc_amd/csrc/xvcgpu.hip -o /tmp/xvcgpu.s
    python tools/isa_mix.py /tmp/xvcgpu.s 'me_search_wave_kernelILi16ELi1ELb0' [...]

Static counts (every instruction once, whatever its loop's trip count): the mix
of a phase instance is applied to that phase's dynamic count from the PMCs
(SQ_INSTS_VALU) to put ONE number on the VALU issue floor.  Rates (clocks per
wave64 instruction per SIMD with 4 waves per SIMD, profiles/r03_valu_rate.txt):
"full" 2.4 - 32-bit add / logic / shift / compare / select / min-max / bfe / add3 /
lshl_add / DPP moves;  "half" 4.3 - v_sad_*, v_dot2*, v_pk_*, v_mul_lo / mul_hi,
v_mad_u32_u24, v_perm, v_alignbit, 16-bit ops;  "wide" 4.5 - v_mad_u64_u32, 64-bit
shifts."""
import collections
import re
import sys

# measured (profiles/r03_valu_rate.txt, 4 waves per SIMD, clocks per wave64 instruction):
#   2.13  the plain VOP1 / VOP2 forms of v_mov, v_add / v_sub(rev)_u32, v_and / v_or / v_xor,
#         v_lshrrev_b32, v_ashrrev_i32, v_sub_u16
#   4.25  everything else that was measured: v_lshlrev_b32 (!), v_min / v_max, v_bfe, v_add3,
#         v_lshl_add, v_mad_u32_u24, v_cndmask (e64), every DPP / SDWA form, v_pk_*, v_dot2*,
#         v_sad_u16, v_perm, v_alignbit, v_mul_lo / _i24, v_lshlrev_b64, v_lshl_add_u64
#   4.56  instructions that write a scalar: v_cmp_*, v_add_co_u32; v_mad_u64_u32
FULL = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_sub_u16", "v_add_u16", "v_not_b32",
        "v_accvgpr", "v_nop")
WIDE = ("v_cmp", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_mad_u64_u32",
        "v_mad_i64_i32", "v_readlane", "v_readfirstlane", "v_writelane")
RATE = {"full": 2.13, "half": 4.25, "wide": 4.56}


def classify(m):
    if m.startswith(WIDE):
        return "wide"
    if m.endswith(("_dpp", "_sdwa", "_e64")):
        return "half"
    base = m[:-4] if m.endswith("_e32") else m
    return "full" if base in FULL else "half"


def kernel_body(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    out = []
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        out.append(l)
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for name in sys.argv[2:]:
        body = kernel_body(lines, name)
        cnt = collections.Counter()
        other = collections.Counter()
        for l in body:
            t = l.strip().split()
            if not t or t[0].startswith((";", ".", "//")) or t[0].endswith(":"):
                continue
            m = t[0]
            if m.startswith("v_"):
                cnt[m] += 1
            else:
                other[m.split("_")[0]] += 1
        by = collections.Counter()
        for m, n in cnt.items():
            by[classify(m)] += n
        total = sum(by.values())
        clk = sum(RATE[k] * n for k, n in by.items()) / max(1, total)
        print("%s: %d VALU instructions (static): full %d (%.0f%%), half %d (%.0f%%), wide %d "
              "(%.0f%%) -> %.2f clocks per VALU instruction; SALU %d, LDS %d, VMEM %d, waitcnt %d" %
              (name, total, by["full"], 100.0 * by["full"] / total, by["half"],
               100.0 * by["half"] / total, by["wide"], 100.0 * by["wide"] / total, clk,
               other["s"] - sum(n for m, n in other.items() if m == "s") + other["s"], other["ds"],
               other["global"] + other["buffer"] + other["flat"] + other["scratch"],
               sum(1 for l in body if "s_waitcnt" in l)))
        print("   top:", ", ".join("%s %d" % kv for kv in cnt.most_common(14)))


if __name__ == "__main__":
    main()
