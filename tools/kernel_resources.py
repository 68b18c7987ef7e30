#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the built library, from the code objects' own metadata
(NT_AMDGPU_METADATA of the gfx950 ELFs inside libmagickhip.so's .hip_fatbin) — no GPU, no disassembly.

    python tools/kernel_resources.py [library.so] [substring ...]

A kernel at the 128-register limit of four waves a SIMD that the compiler pushes over it SPILLS: a scratch reload
inside a walk waits for every store in flight (s_waitcnt vmcnt(0)) and cost the exact fused blur 6 % in round 6 before
the listing was read.  tests/test_kernel_resources.py pins the hot kernels to zero scratch."""
import os
import struct
import subprocess
import sys
import tempfile

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIBRARY = os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so")
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def fat_binary(library):
    with tempfile.TemporaryDirectory() as scratch:
        out = os.path.join(scratch, "fat.bin")
        subprocess.run([OBJCOPY, "--dump-section", ".hip_fatbin=" + out, library], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(out, "rb") as f:
            return f.read()


def device_objects(blob):
    """The gfx950 ELFs of every bundle (one bundle per translation unit)."""
    at = blob.find(MAGIC)
    while at >= 0:
        (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        cursor = at + len(MAGIC) + 8
        for _ in range(count):
            offset, size, triple_len = struct.unpack_from("<QQQ", blob, cursor)
            triple = blob[cursor + 24:cursor + 24 + triple_len].decode()
            cursor += 24 + triple_len
            if "amdgcn" in triple and size:
                yield blob[at + offset:at + offset + size]
        at = blob.find(MAGIC, at + 1)


def kernels_of(elf):
    """[{name, vgprs, sgprs, scratch, lds}] out of an AMDGPU ELF's metadata note."""
    if elf[:4] != b"\x7fELF":
        return []
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    out = []
    for i in range(shnum):
        base = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, base + 4)
        if sh_type != 7:                              # SHT_NOTE
            continue
        offset, size = struct.unpack_from("<QQ", elf, base + 0x18)
        cursor, end = offset, offset + size
        while cursor + 12 <= end:
            namesz, descsz, kind = struct.unpack_from("<III", elf, cursor)
            name = elf[cursor + 12:cursor + 12 + namesz].rstrip(b"\0")
            desc_at = cursor + 12 + ((namesz + 3) & ~3)
            if name == b"AMDGPU" and kind == 32:      # NT_AMDGPU_METADATA
                meta = msgpack.unpackb(elf[desc_at:desc_at + descsz], raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    out.append({"symbol": k.get(".name", ""), "vgprs": k.get(".vgpr_count", 0) + k.get(".agpr_count", 0),
                                "sgprs": k.get(".sgpr_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                                "lds": k.get(".group_segment_fixed_size", 0)})
            cursor = desc_at + ((descsz + 3) & ~3)
    return out


def demangled(symbols):
    try:
        text = subprocess.run(["c++filt"], input="\n".join(symbols), capture_output=True, text=True, check=True).stdout
        names = text.splitlines()
        if len(names) == len(symbols):
            return names
    except Exception:
        pass
    return list(symbols)


def kernel_resources(library=DEFAULT_LIBRARY):
    rows = []
    for elf in device_objects(fat_binary(library)):
        rows.extend(kernels_of(elf))
    for row, name in zip(rows, demangled([r["symbol"] for r in rows])):
        row["name"] = name
    return rows


def main():
    args = sys.argv[1:]
    library = args.pop(0) if args and args[0].endswith(".so") else DEFAULT_LIBRARY
    rows = kernel_resources(library)
    rows = [r for r in rows if not args or any(a in r["name"] for a in args)]
    for r in sorted(rows, key=lambda r: (-r["scratch"], r["name"])):
        print("%4d vgprs %4d sgprs %6d B scratch %6d B lds  %s" % (r["vgprs"], r["sgprs"], r["scratch"], r["lds"], r["name"][:150]))
    print("%d kernels, %d with scratch" % (len(rows), sum(1 for r in rows if r["scratch"])))


if __name__ == "__main__":
    main()
