#!/usr/bin/env python3
"""sha256 over the MACHINE CODE of the library's gfx950 kernels (.text and .rodata — the instructions and the kernel descriptors — of the
code object inside libspades_mi355x.so). A PMC traffic table under profiles/ is a measurement of kernels: it stays valid while this hash
is the one it was taken on, whatever happened to the host code around them (bench.py, tools/pmc_summary.py). Symbol tables and notes are
left out on purpose: clang derives internal symbol suffixes from a hash of the translation unit, so they change with any host-side edit.
usage: devcode_hash.py [path/to/libspades_mi355x.so]"""
import hashlib
import os
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_code_hash(lib=None):
    """-> 16 hex digits, or None when the tools or the library are missing"""
    lib = lib or os.path.join(ROOT, "spades_amd", "csrc", "libspades_mi355x.so")
    try:
        with tempfile.TemporaryDirectory() as td:
            fb, co = os.path.join(td, "fatbin"), os.path.join(td, "co.elf")
            subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb], stderr=subprocess.DEVNULL)
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}",
                                   f"--output={co}", "--unbundle"], stderr=subprocess.DEVNULL)
            h = hashlib.sha256()
            for sec in (".text", ".rodata"):
                out = os.path.join(td, sec[1:])
                subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", f"--only-section={sec}", co, out], stderr=subprocess.DEVNULL)
                data = open(out, "rb").read()
                if not data:
                    return None
                h.update(data)
            return h.hexdigest()[:16]
    except (OSError, subprocess.CalledProcessError):
        return None


if __name__ == "__main__":
    print(device_code_hash(sys.argv[1] if len(sys.argv) > 1 else None))
