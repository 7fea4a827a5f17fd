#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc --save-temps output (the gfx950 .s file).

  tools/isa_mix.py <file.s> <substring of the mangled kernel name> [--loops]

Prints the kernel's VGPR / SGPR / LDS / scratch use and its instructions by class (VALU, 64-bit VALU, `valu_quarter` = 32-bit
multiplies and 64-bit mads -- counted on their own, but full rate on gfx950: tools/ubench_valu.hip --, LDS, global/flat, SALU,
branches, waits), for the whole body and per basic block with --loops.  A VALU-bound
kernel's time goes with the weighted VALU count of its hot loop, which is what this is for: iterating on instruction count
without a GPU."""
import re
import sys

# (32-bit integer multiplies are FULL rate on gfx950 -- tools/ubench_valu.hip, round 4 --; the class is kept as a count, weighted 1)
QUARTER = re.compile(r"^v_(mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_lo_i32|mad_u64_u32|mad_i64_i32|mul_u64|mul_f64|fma_f64|add_f64|rcp_f64|div)")
HALF64 = re.compile(r"^v_(lshlrev_b64|lshrrev_b64|ashrrev_i64|cmp_[a-z]+_[ui]64|cmpx_[a-z]+_[ui]64)")


def classify(op):
    if op.startswith("v_"):
        if QUARTER.match(op):
            return "valu_quarter"
        if HALF64.match(op):
            return "valu_64"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    per_block = "--loops" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w$.]*):", l)
        if m and pat in m.group(1):
            start = i
            name = m.group(1)
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    tot = {}
    blocks = []
    cur = ["entry", {}]
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if s.startswith(".LBB") and ":" in s:
                s = s.split(":")[0] + ":"
                blocks.append(cur)
                cur = [s[:-1], {}]
            continue
        op = s.split()[0]
        c = classify(op)
        tot[c] = tot.get(c, 0) + 1
        cur[1][c] = cur[1].get(c, 0) + 1
        if c in ("valu_quarter",):
            cur[1].setdefault("_q", []).append(op)
    blocks.append(cur)
    meta = {}
    for l in lines[end:end + 200]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\S+)", l)
        if m:
            meta[m.group(1)] = m.group(2)
        if l.startswith("\t.end_amdhsa_kernel"):
            break
    print(name)
    print("  ", meta)
    w = lambda d: d.get("valu", 0) + 2 * d.get("valu_64", 0) + d.get("valu_quarter", 0)
    print("   total:", dict(sorted(tot.items())), "weighted VALU:", w(tot))
    if per_block:
        for nm, d in blocks:
            n = sum(v for k, v in d.items() if not k.startswith("_"))
            if n >= 12:
                print("   %-12s n=%4d wVALU=%4d %s" % (nm, n, w(d), {k: v for k, v in sorted(d.items()) if not k.startswith("_")}))


if __name__ == "__main__":
    main()
