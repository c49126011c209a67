#!/usr/bin/env python3
"""The issue cost of a kernel's instruction MIX: static VALU opcode histogram of one kernel (hipcc -S of its source)
priced with the per-opcode issue intervals tools/ubench/valu_rate.hip measured on the chip.

  python tools/valu_mix.py profiles/r04_valu_rate_ubench.txt [kernel-substring=ped_accumulate_kernel] [source=pedersen.hip]

Prints one JSON object: the histogram, the interval of every opcode class at 2 and at 8 waves per SIMD, and the
mix-weighted interval  c_mix = sum_i share_i * cycles_i  - the denominator of the VALU-issue roofline
(peak = 1024 SIMDs x clock / c_mix wave64 instructions per second).  Static counts: the hot loop of the hash kernels
is straight-line code, prologue and epilogue have the same mix (they are the same inlined fe_mul / fe_sqr)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ubench_table(path):
    rows = collections.defaultdict(dict)
    pat = re.compile(r"(.+?)\s+chains=\s*(\d+) waves/SIMD=(\d)\s+[\d.]+ ms/launch\s+clock\s+(\d+) MHz\s+([\d.]+) cycles/instr/SIMD")
    for line in open(path):
        m = pat.match(line)
        if m and int(m.group(2)) == 16:
            rows[m.group(1).strip()][int(m.group(3))] = float(m.group(5))
    return rows


# opcode (without _e32 / _e64 / _dpp suffix) -> row of the micro-benchmark that stands for it
CLASS = {
    "v_mad_i64_i32": "v_mad_i64_i32 (sgpr carry)", "v_mad_u64_u32": "v_mad_u64_u32 (vcc)",
    "v_and_b32": "v_and_b32", "v_or_b32": "v_and_b32", "v_xor_b32": "v_xor_b32", "v_not_b32": "v_and_b32",
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_sub_u32", "v_mov_b32": "v_mov_b32",
    "v_ashrrev_i32": "v_ashrrev_i32", "v_lshlrev_b32": "v_lshlrev_b32", "v_lshrrev_b32": "v_lshlrev_b32",
    "v_lshl_add_u64": "v_lshl_add_u64", "v_ashrrev_i64": "v_ashrrev_i64", "v_lshlrev_b64": "v_lshlrev_b64",
    "v_lshrrev_b64": "v_lshrrev_b64", "v_mov_b64": "v_mov_b64", "v_alignbit_b32": "v_alignbit_b32",
    "v_lshl_add_u32": "v_lshl_add_u32", "v_lshl_or_b32": "v_lshl_or_b32", "v_add3_u32": "v_add3_u32",
    "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_i32", "v_mul_lo_u32": "v_mul_lo_u32", "v_mul_hi_u32": "v_mul_hi_u32",
    "v_add_co_u32": "v_add_co_u32", "v_addc_co_u32": "v_addc_co_u32", "v_sub_co_u32": "v_sub_co_u32",
    "v_subb_co_u32": "v_addc_co_u32", "v_subrev_co_u32": "v_sub_co_u32", "v_cndmask_b32": "v_cndmask_b32 (sgpr)",
    "v_cmp_eq_u32": "v_cmp_ne_u32", "v_cmp_ne_u32": "v_cmp_ne_u32", "v_cmp_gt_u32": "v_cmp_ne_u32",
    "v_cmp_lt_u32": "v_cmp_ne_u32", "v_cmp_gt_i32": "v_cmp_ne_u32", "v_cmp_lt_i32": "v_cmp_ne_u32",
    "v_fma_f64": "v_fma_f64", "v_mul_f64": "v_mul_f64", "v_add_f64": "v_add_f64", "v_rcp_f64": "v_rcp_f64",
    "v_rndne_f64": "v_rndne_f64", "v_cvt_f64_i32": "v_cvt_f64_i32", "v_cvt_i32_f64": "v_cvt_i32_f64",
}
DEFAULT_ROW = "v_lshl_or_b32"  # anything not listed: priced as a three-operand 32-bit instruction (the 4-cycle class)


def main():
    table = ubench_table(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 else "ped_accumulate_kernel"
    src = sys.argv[3] if len(sys.argv) > 3 else "pedersen.hip"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               os.path.join(ROOT, "stark-perpetual_amd", "csrc", src), "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = end = name = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if start is None and m and want in l:
            start, name = i, m.group(1)
        elif start is not None and ".amdhsa_kernel " + name in l:
            end = i
            break
    hist = collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if m:
            hist[re.sub(r"_(e32|e64|dpp|sdwa)$", "", m.group(1))] += 1
    total = sum(hist.values())
    res = {"kernel": name, "source": src, "static_valu_instructions": total, "ubench": os.path.basename(sys.argv[1]),
           "histogram": dict(hist.most_common()), "priced_as": {}, "unlisted_opcodes": []}
    for waves in (2, 8):
        cyc = 0.0
        for op, n in hist.items():
            row = CLASS.get(op)
            if row is None:
                row = DEFAULT_ROW
                if op not in res["unlisted_opcodes"]:
                    res["unlisted_opcodes"].append(op)
            cyc += n * table[row][waves]
            res["priced_as"][op] = {"row": row, "cycles_at_2_waves": table[row][2], "cycles_at_8_waves": table[row][8]}
        res["cycles_per_instr_mix_at_%d_waves_per_simd" % waves] = round(cyc / total, 3)
    # the floor: every opcode at the best interval it reached at any occupancy
    best = sum(n * min(table[CLASS.get(op, DEFAULT_ROW)].values()) for op, n in hist.items()) / total
    res["cycles_per_instr_mix_best_of_any_occupancy"] = round(best, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
