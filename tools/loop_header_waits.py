#!/usr/bin/env python3
"""List, per gfx950 kernel of a .hip translation unit, the inner-loop headers that open with `s_waitcnt vmcnt(0)`.

    python tools/loop_header_waits.py pixel-nerf_amd/csrc/pnr_split.hip [--filter eval_split] [-DMACRO ...]

Why: a register spill that hipcc reloads right in front of a GEMM loop shares the vmcnt counter with the weight ring's
requests; the loop header is one instruction for both predecessors, so it becomes `s_waitcnt vmcnt(0)` and the ring (4 k-steps of
prefetch) drains at every loop body.  tools/kernel_regs.py's spill COUNT does not show that; this does (round 6: found in a
build with 40 spills whose A/B gain had vanished; profiles/r06_split_kernel_ab.txt)."""
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    flt = None
    extra = []
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == "--filter":
            flt = args.pop(0)
        else:
            extra.append(a)
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S",
                        src, "-o", f.name] + extra, check=True, stderr=subprocess.DEVNULL)
        lines = open(f.name).read().split("\n")
    name, loops, bad = None, 0, 0
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, loops, bad = m.group(1), 0, 0
        if "Inner Loop Header" in ln and name:
            loops += 1
            if any("vmcnt(0)" in x for x in lines[i + 1:i + 4]):
                bad += 1
        if "s_endpgm" in ln and name:
            out.append((name, loops, bad))
            name = None
    dem = subprocess.run(["c++filt"] + [n for n, _, _ in out], capture_output=True, text=True).stdout.split("\n")
    print("loops  vmcnt(0)-headers  kernel")
    for (n, l, b), d in zip(out, dem):
        if flt and flt not in d:
            continue
        print(f"{l:5d}  {b:16d}  {d}")


if __name__ == "__main__":
    main()
