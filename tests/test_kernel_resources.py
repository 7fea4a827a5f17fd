"""What the compiler made of the hot kernels, read from the gfx950 code objects inside the built objects (kat_amd/build/*.o): no kernel
of the library spills to scratch, and the stage kernels keep the register / LDS budgets their launch shapes are designed around
(three level-1 workgroups per CU, two 512-thread or one 1024-thread apply workgroup, ...).  An occupancy
regression shows up here on the CPU instead of as a slower bench on the GPU."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(tmp_path):
    objs = sorted(glob.glob(os.path.join(ROOT, "kat_amd", "build", "kg_*.o")))
    if not objs or not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")) or not shutil.which("c++filt"):
        pytest.skip("no built objects / no LLVM tools")
    out = {}
    for o in objs:
        fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, o], capture_output=True, text=True)
        if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            continue                                                      # a host-only object
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        os.remove(fat)
        for blk in notes.split("- .agpr_count")[1:]:
            f = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)
            name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name).replace("void ", "").replace("kg::", "")
            out[name] = {"vgpr": int(f("vgpr_count")), "scratch": int(f("private_segment_fixed_size")), "lds": int(f("group_segment_fixed_size")),
                         "vgpr_spill": int(f("vgpr_spill_count")), "wg": int(f("max_flat_workgroup_size"))}
    return out


def test_no_scratch_and_the_stage_kernels_keep_their_budgets(tmp_path):
    ks = kernels(tmp_path)
    assert len(ks) > 100
    # No kernel of the library spills -- but for two shapes the register allocator lands one step past its budget on: the exact level 2 --
    # the fall-back of the one-pass edition -- a handful in its item-by-item copy-out, and the one-pass level 2 of 6-byte items from 6-byte
    # items (remainders of 40-47 bits: not the bench's shape) four loop-invariant values.
    # Round 6: the block edition of the one-pass level 2 (k_p2_fast<1, false, ...>) keeps the next tile's 24 registers of loads in flight across
    # its phases and sits exactly on its 128-register budget: ONE loop-invariant pointer pair is spilled before the bucket loop and reloaded once per
    # bucket (1479 tiles at the bench's size) -- the tile loop itself touches no scratch (read from the ISA: the reload sits at loop depth 1).
    # Its form that reads level 1's blocks of ten (the last template argument) spills six such pairs -- the per-lane addresses of a bucket's first
    # tile, again before the tile loop and reloaded once per bucket.
    # kg_l2_blocks.hpp's kernel (the bench's shape since round 6) spills four address pairs of a bucket's FIRST tile request, before the tile loop
    # (twice per workgroup and pass); inside the loop the requests are base + 32-bit offset and nothing touches scratch (read from the ISA).
    allowed = {"k_p2<": 40, "k_p2_fast<2, false": 24, "k_p2_fast<1, false": 16, "k_p2_fast<1, false, false, true": 64, "k_p2_fast<1, false, true, true": 64,
               "k_p2x_fast<": 40}
    spilled = {n: v for n, v in ks.items() if v["scratch"] > max([lim for pre, lim in allowed.items() if n.startswith(pre)], default=0)}
    assert not spilled, spilled

    def every(prefix, **limits):
        hit = {n: v for n, v in ks.items() if n.startswith(prefix)}
        assert hit, prefix
        for n, v in hit.items():
            for key, lim in limits.items():
                assert v[key] <= lim, (n, key, v[key], lim)
    # level 1: three 512-thread workgroups per CU = six waves per SIMD (512 / 6 = 85 VGPRs) and 3 x 42 LDS granules of 1280 bytes
    # (the segmented editions of tables with at most 512 level-1 digits, which every round of size takes; the others -- more digits,
    # or the exact edition of small rounds and skewed inputs -- run two per CU)
    every("k_p1v2_scatter<true, true, 512>", vgpr=84, lds=42 * 1280)
    every("k_p1v2_scatter<true, false, 512>", vgpr=84, lds=42 * 1280)
    every("k_p1v2_scatter<", vgpr=128, lds=62 * 1280)
    # level 1, block edition: ONE 1024-thread workgroup per CU = four waves per SIMD (its 158 KB of LDS are dynamic: kg_l1_blocks.hpp asserts them).
    # Not a byte of scratch: a reload anywhere in its tile loop would wait on the vector-memory counter, i.e. for the stores the loop spreads out.
    every("k_p1b_scatter<", vgpr=128)
    assert all(v["scratch"] == 0 for n, v in ks.items() if n.startswith("k_p1b_scatter<")), {n: v for n, v in ks.items() if n.startswith("k_p1b_scatter<")}
    # level 2: one 1024-thread workgroup per CU = four waves per SIMD
    every("k_p2_fast<", vgpr=128)
    every("k_p2x_fast<", vgpr=128)
    every("k_p2<", vgpr=128)
    # the applies: four waves per SIMD (two 512-thread workgroups, or one of 1024 threads)
    every("k_p3_apply_pk<", vgpr=128)
    every("k_p3_apply2<", vgpr=128)
    # the packed shape the bench runs must leave room for its partner on the CU
    every("k_p3_apply_pk<512, 10, 1, false, false, false, 2, false, 1>", vgpr=128, lds=64)
    every("k_comp_fused<", vgpr=128)           # one 1024-thread workgroup per CU: all eight shapes (resident table, pairs per lane, side-table counts)
    assert len([n for n in ks if n.startswith("k_comp_fused<")]) == 8
    every("k_merge_apply<", vgpr=128)
    every("k_w3_apply", vgpr=128)
    every("k_w1<", vgpr=128, lds=16 * 1024)
    # the streaming reducers run two 1024-thread workgroups per CU (hist) or one beside a 108 KB matrix (gcp)
    every("k_hist<", vgpr=64)
    every("k_gcp_pk", vgpr=64)
    every("k_gcp<", vgpr=64)
