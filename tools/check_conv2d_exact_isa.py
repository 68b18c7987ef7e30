#!/usr/bin/env python3
"""The product loops of convolve2d_exact.hip read their operands through asm and wait by count, so
between a kernel's first and last v_mfma there must be no copy or spill of a vector register (to
the compiler an asm's result exists from the asm on) and no scalar load (SMEM shares lgkmcnt with
the LDS and returns out of order).  Builds the ISA with the flags of the library and checks.
   python tools/check_conv2d_exact_isa.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["make", "-C", os.path.join(ROOT, "imagemagick_amd", "csrc"), "asm", "FILE=convolve2d_exact"],
               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
path = os.path.join(ROOT, "imagemagick_amd", "build", "convolve2d_exact.s")
text = open(path).read()
bad = 0
for name, body in re.findall(r"^(_ZN2mh19conv2d_exact_kernel\w+):.*?\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    lines = body.split("\n")
    mfma = [i for i, l in enumerate(lines) if "v_mfma" in l]
    loop = lines[mfma[0]:mfma[-1] + 1]
    # registers the asm reads fill
    filled = set()
    for l in loop:
        m = re.search(r"ds_read_b128 v\[(\d+):(\d+)\]", l)
        if m:
            filled.update(range(int(m.group(1)), int(m.group(2)) + 1))

    def touches(l):
        # a SOURCE operand among them (an address moved into one before its own read is fine)
        operands = l.split(",", 1)[1] if "," in l else ""
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", operands):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(r) for r in re.findall(r"\bv(\d+)\b", operands))
        return bool(regs & filled)
    suspicious = [l.strip() for l in loop if re.search(r"\b(scratch_|s_load|s_buffer_load)", l)
                  or (re.search(r"\b(v_mov_b32|v_mov_b64|v_accvgpr|v_readlane|v_writelane)", l) and touches(l))]
    print("%-60s %3d products in the loop, %d suspicious" % (name, len(mfma), len(suspicious)))
    for l in suspicious[:6]:
        print("     ", l)
    bad += len(suspicious)
os.remove(path)
sys.exit(1 if bad else 0)
