#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel inside libecc_hip.so, read from the code objects' own metadata (no GPU, no
ROCm tool): the .so carries one clang offload bundle per translation unit, each with an AMDGPU ELF whose NT_AMDGPU_METADATA note
is a msgpack map ('amdhsa.kernels').

usage: scripts/kernel_resources.py [libecc_hip.so]     -- prints one line per kernel

tests/test_kernel_resources.py pins the numbers the launch design depends on (the pair kernel's seven waves per SIMD, the wide
k01 kernels fitting the hole a retiring pair-kernel workgroup leaves)."""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """The gfx950 ELF images of every offload bundle in `blob`."""
    at = blob.find(MAGIC)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        o = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tlen].decode()
            o += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + 1)


def elf_notes(elf):
    """(name, type, desc) of every note of a little-endian ELF64 image."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1
    e_phoff, = struct.unpack_from("<Q", elf, 0x20)
    e_phentsize, e_phnum = struct.unpack_from("<HH", elf, 0x36)
    for i in range(e_phnum):
        p_type, _flags, p_offset, _va, _pa, p_filesz = struct.unpack_from("<IIQQQQ", elf, e_phoff + i * e_phentsize)
        if p_type != 4:  # PT_NOTE
            continue
        o, end = p_offset, p_offset + p_filesz
        while o + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, o)
            o += 12
            name = elf[o:o + namesz].rstrip(b"\0").decode()
            o += (namesz + 3) & ~3
            desc = elf[o:o + descsz]
            o += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(path):
    """{demangled-ish kernel symbol: metadata dict} over all code objects of the library."""
    out = {}
    blob = open(path, "rb").read()
    for elf in code_objects(blob):
        for name, ntype, desc in elf_notes(elf):
            if name == "AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out


def find(ks, *parts):
    """The kernels whose mangled name contains every one of `parts`."""
    return {n: k for n, k in ks.items() if all(p in n for p in parts)}


if __name__ == "__main__":
    import os
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "epipolarconsistency_amd", "libecc_hip.so")
    ks = kernels(lib)
    print("%-86s %5s %5s %8s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "lds"))
    for n in sorted(ks):
        k = ks[n]
        print("%-86s %5d %5d %8d %7d" % (n[:86], k[".vgpr_count"], k[".sgpr_count"], k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))
