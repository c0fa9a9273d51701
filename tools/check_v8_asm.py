#!/usr/bin/env python3
"""Mechanical check of the invariant behind the uncounted loads of k_render_v8 (rt_kernels.hip) and k_wf_trace
(rt_wavefront.hip, same joined pair / leaf step).

The kernel issues its pair / triangle loads in inline assembly that the compiler does not count; the matching
`s_waitcnt vmcnt(0)` sits in separate asm statements at the head of the two consuming sections.  That is only
correct while the compiler places no instruction that reads or writes the four destination register tuples between
the load group and those waits (a copy, a spill), and while the kernel has no register spills at all (a spill store
of such a register would read it in flight).  This script reads the device assembly (hipcc -S) and fails if

  * a spill or reload (scratch_store / scratch_load ... Folded Spill / Reload) sits between a load group and its
    consumers' waits (elsewhere spills are harmless to the invariant and only reported: the persistent mode of
    k_wf_trace spills in its shade phase, outside the step loop), or
  * on any straight-line stretch between RT_V8_LOADS_END and the RT_V8_WAIT_* markers an instruction names a
    destination register of the load group before an `s_waitcnt vmcnt(0)` has been passed in that basic block.

Usage: check_v8_asm.py rt_kernels.s [rt_wavefront.s ...]   (exit status 1 on a violation)."""
import re
import sys


def regs_of(token):
    """v[10:13] -> {10..13}; v7 -> {7}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", token)
    return {int(m.group(1))} if m else set()


def vregs_in(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line.split(";")[0]):
        out |= regs_of(tok)
    return out


def check(path):
    lines = open(path).read().split("\n")
    errors, groups = [], 0
    kernel = None
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w*(?:k_render_v8|k_wf_trace)\w*):", ln)
        if m:
            kernel = m.group(1)
        if "RT_V8_LOADS_BEGIN" in ln:
            groups += 1
            dest = set()
            lds_reads = False  # the group also reads LDS: the consumers' waits must cover lgkmcnt too
            j = i + 1
            while "RT_V8_LOADS_END" not in lines[j]:
                mm = re.match(r"\s*(?:global_load_dwordx\d|ds_read_b\d+)\s+(\S+),", lines[j])  # (k_wf_trace: LDS copies of the top records)
                if mm:
                    dest |= regs_of(mm.group(1))
                    lds_reads |= "ds_read" in lines[j]
                j += 1
            if len(dest) != 14:
                errors.append(f"{kernel}: load group at line {i + 1} defines {len(dest)} registers, expected 14")
            # scan forward until both consumer waits have been seen
            seen = set()
            waited = False  # within the current basic block
            k = j + 1
            while k < len(lines) and len(seen) < 2:
                t = lines[k]
                code = t.split(";")[0].strip()
                if re.match(r"^\.?\w+:", t.strip()):      # a label starts a new basic block
                    waited = False
                if "RT_V8_WAIT_TRI" in t:
                    seen.add("tri")
                if "RT_V8_WAIT_PAIR" in t:
                    seen.add("pair")
                if ("RT_V8_WAIT_TRI" in t or "RT_V8_WAIT_PAIR" in t) and lds_reads and "lgkmcnt(0)" not in code:
                    errors.append(f"{kernel}: line {k + 1}: the load group reads LDS too, but this consumer's wait leaves the "
                                  f"in-flight ds_read results uncovered (no lgkmcnt(0)): {code}")
                if code.startswith("s_waitcnt") and "vmcnt(0)" in code and (not lds_reads or "lgkmcnt(0)" in code):
                    waited = True
                elif code and not code.startswith(".") and not waited and (vregs_in(code) & dest):
                    errors.append(f"{kernel}: line {k + 1} touches a register of the in-flight load group "
                                  f"(v{sorted(vregs_in(code) & dest)}) before s_waitcnt vmcnt(0): {code}")
                if re.match(r"scratch_(load|store)", code) and ("Spill" in t or "Reload" in t):
                    errors.append(f"{kernel}: line {k + 1}: register spill traffic between a load group and its consumers' waits: {code}")
                if code.startswith("s_endpgm"):
                    errors.append(f"{kernel}: reached s_endpgm before both RT_V8_WAIT markers")
                    break
                k += 1
            i = j
        i += 1
    # register spills, from the kernel descriptors' comment block / metadata
    for m in re.finditer(r"\.name:\s+(_Z\w*(?:k_render_v8|k_wf_trace)\w*)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)",
                         "\n".join(lines)):
        if int(m.group(3)):  # (checked above: none of them between a load group and its waits)
            print(f"check_v8_asm: note: {m.group(1)[:60]}... has {m.group(3)} vector register spills, none inside a load window")
        elif int(m.group(2)):  # scalar spills go to lanes of a reserved VGPR (v_writelane), never to memory: no hazard
            print(f"check_v8_asm: note: {m.group(1)[:40]}... has {m.group(2)} scalar register spills (to VGPR lanes)")
    if groups == 0:
        errors.append("no RT_V8_LOADS_BEGIN marker found: the check did not see the kernel")
    return groups, errors


if __name__ == "__main__":
    bad = 0
    for path in sys.argv[1:]:
        n, errs = check(path)
        for e in errs:
            print("check_v8_asm: " + e, file=sys.stderr)
        print(f"check_v8_asm: {path}: {n} load group(s) checked, {len(errs)} violation(s)")
        bad += len(errs)
    sys.exit(1 if bad else 0)
