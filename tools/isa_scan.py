#!/usr/bin/env python3
"""Disassemble the gfx950 code objects embedded in libvvae_hip.so and count instructions by regex (no GPU needed).

    python tools/isa_scan.py 'v_pk_[a-z0-9]+_f32' 'ds_bpermute'

The library's .hip_fatbin section is a sequence of clang offload bundles (one per .hip file); each holds a host placeholder and one
amdgcn-amd-amdhsa--gfx950 ELF, which llvm-objdump disassembles.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "video_vae_amd", "libvvae_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB):
    """-> list of gfx950 ELF images (bytes) found in the shared library."""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        base = blob.find(MAGIC, pos)
        if base < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tsize].decode()
            off += 24 + tsize
            if "gfx950" in triple and esize:
                out.append(blob[base + eoff:base + eoff + esize])
        pos = base + len(MAGIC)
    return out


def disassemble(path=LIB):
    text = []
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            text.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout)
    return "\n".join(text)


def count(patterns, path=LIB):
    asm = disassemble(path)
    return {p: len(re.findall(r"^\s+" + p + r"\b", asm, flags=re.M)) for p in patterns}, asm.count("\n")


if __name__ == "__main__":
    pats = sys.argv[1:] or [r"v_pk_[a-z0-9]+_f32", r"ds_bpermute_b32", r"v_mfma_\w+"]
    res, lines = count(pats)
    print(f"{len(code_objects())} code objects, {lines} lines of disassembly")
    for p, n in res.items():
        print(f"{n:8d}  {p}")
