#!/usr/bin/env python3
"""Kernel resource table from the device assembly the build writes (raytracing_amd/csrc/*.gfx950.s, `make`):
VGPRs, SGPRs, LDS per workgroup, scratch per lane, spills and the waves per SIMD those allow, one row per kernel
instance, read from the amdhsa.kernels metadata — the same numbers `make usage` prints as compiler remarks and
rocprofv3's kernel trace reports per dispatch, read as the footnotes printed under the table say.
Usage: usage_table.py FILE.s [FILE.s ...] > profiles/rNN_x/resource_usage.txt"""
import re, subprocess, sys

CXXFILT = "c++filt"


def waves_per_simd(vgprs, lds, block):
    # gfx950: 512 VGPRs per lane per SIMD, allocated in blocks of 8; at most 8 waves per SIMD; 160 KiB LDS per CU
    alloc = max(8, (vgprs + 7) // 8 * 8)
    w = min(8, 512 // alloc)
    if lds:
        blocks = (160 * 1024) // lds
        w = min(w, blocks * (block // 64) // 4)
    return w


rows = []
for path in sys.argv[1:]:
    text = open(path).read()
    meta = text[text.index("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        def g(key, d=0):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else d
        name = g("name")
        dem = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"^void ", "", dem)
        dem = dem[: dem.index("(")] if "(" in dem else dem
        block = int(g("max_flat_workgroup_size", 256))
        v, s, lds, scr = int(g("vgpr_count")), int(g("sgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size"))
        rows.append((dem, v, s, lds, scr, int(g("vgpr_spill_count")), int(g("sgpr_spill_count")), block, waves_per_simd(v, lds, block)))
w = max(len(r[0]) for r in rows)
print(f"{'kernel':<{w}}  VGPR  SGPR  LDS B/wg  scratch B/lane  vspill  sspill  max wg  waves/SIMD")
for r in rows:
    print(f"{r[0]:<{w}}  {r[1]:>4}  {r[2]:>4}  {r[3]:>8}  {r[4]:>14}  {r[5]:>6}  {r[6]:>6}  {r[7]:>6}  {r[8]:>10}")
print("""
Notes
 * LDS B/wg is the STATIC size.  The traversal kernels take their stack heads as DYNAMIC LDS at launch
   (NLDS x 256 lanes x 4 B): k_trace_closest / k_render_v0 declare theirs statically (24 KiB, above), k_render_v8 gets
   24 KiB per launch, k_wf_trace<..,16,false> 16 KiB, k_wf_trace<..,12,true> 12 KiB + 4 x 1040 B staging.  None of these
   lowers the waves/SIMD column (160 KiB per CU).
 * scratch B/lane of the traversal kernels is the tail of the traversal stack (RT_MAX_STACK - NLDS entries), touched
   only by the rare lane deeper than NLDS; vspill = 0 everywhere: nothing else lives in scratch.
 * rocprofv3 --kernel-trace on gfx950 prints VGPR_Count = allocated VGPRs / 2 (77 -> 80 allocated -> "40";
   110 -> 112 -> "56"), SGPR_Count rounded up to 16, and LDS_Block_Size = 0 for dynamic LDS: the trace and this table
   agree once read that way (profiles/r02_final/kernel_trace.csv: k_wf_trace 40 / 80 / scratch 464, k_render_v8 56 / 112 / 432).""")
