#!/usr/bin/env python3
"""Static check of the closed-form 3-D kernel's inline-asm LDS reads (diffnet_amd/csrc/poisson3d_q1_cf.hip).

The gather's ds_read2_b32 are inline asm and their waits are written by hand (the compiler does not count asm LDS accesses).  This script
disassembles every instantiation and verifies, for every ds_read* in it, that no instruction reads the destination registers before an
`s_waitcnt lgkmcnt(k)` that covers the read: LDS accesses complete in order, so the read has landed once at most k LGKM operations are
outstanding and at least k DS instructions were issued after it.  It also reports a copy (v_mov / v_pk_mov / spill) of a destination
register ahead of its wait -- the failure mode of carrying such a read over a loop's back edge.

usage: python tools/check_lds_waits.py [extra hipcc flags]      (exit status 1 on a violation)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diffnet_amd", "csrc", "poisson3d_q1_cf.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "--offload-arch=gfx950", "-mllvm", "-amdgpu-sdwa-peephole=0"]


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    ops = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", parts[1])]
    ops = [re.sub(r"\s+(offset\d?:\S+|neg_\w+:\S+|op_sel\w*:\S+|sc\d|nt|off)$", "", o).split()[0] if o else o for o in ops]
    return parts[0], ops


def check_function(name, lines):
    bad = []
    n = len(lines)
    for i, line in enumerate(lines):
        op, ops = operands(line)
        if not op.startswith("ds_read") or not ops:
            continue
        dst = regs(ops[0])
        if not dst:
            continue
        ds_after = 0
        covered = False
        for j in range(i + 1, min(n, i + 4000)):
            l2 = lines[j]
            op2, ops2 = operands(l2)
            if not op2 or op2.endswith(":") or op2.startswith("."):
                if op2.endswith(":"):
                    break          # a label: control flow joins here -- the straight-line argument ends (reads are waited for inside their block)
                continue
            if op2 == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", l2)
                if m and int(m.group(1)) <= ds_after:
                    covered = True
                    break
                continue
            if op2.startswith("s_cbranch") or op2 == "s_branch" or op2 == "s_endpgm":
                break
            srcs = set()
            is_store_like = op2.startswith(("ds_write", "global_store", "scratch_store", "ds_bpermute", "flat_store", "buffer_store"))
            for k, t in enumerate(ops2):
                if k == 0 and not is_store_like:
                    continue
                srcs |= regs(t)
            if srcs & dst:
                bad.append((i, j, line.strip(), l2.strip()))
                break
            if not is_store_like and ops2 and regs(ops2[0]) & dst:
                break              # overwritten before any use: the read is dead, nothing to wait for
            if op2.startswith("ds_"):
                ds_after += 1
        _ = covered
    return bad


def main():
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-S", "--cuda-device-only", "-o", "-", SRC], check=True, capture_output=True,
                         text=True).stdout.split("\n")
    funcs, cur, name = {}, None, None
    for line in asm:
        m = re.match(r"^(_ZN2dn22poisson3d_q1_cf_kernel\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                funcs[name] = cur
                cur = None
    total = 0
    for name, lines in sorted(funcs.items()):
        bad = check_function(name, lines)
        nread = sum(1 for l in lines if l.strip().startswith("ds_read"))
        fl = re.search(r"ILi(\d+)E", name).group(1)
        print(f"FL={fl:>3}: {nread:3d} LDS reads, {len(bad)} used before a covering wait")
        for i, j, a, b in bad[:4]:
            print(f"    line {i}: {a}\n    line {j}: {b}")
        total += len(bad)
    print("instantiations:", len(funcs), " violations:", total)
    sys.exit(1 if total or not funcs else 0)


if __name__ == "__main__":
    main()
