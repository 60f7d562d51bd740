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


def _code_object(lib, td):
    fb, co = os.path.join(td, "fatbin"), os.path.join(td, "co.elf")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb], stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}",
                           f"--output={co}", "--unbundle"], stderr=subprocess.DEVNULL)
    return co


def kernel_code_hash(names, lib=None):
    """sha256 over the machine code and the kernel descriptors of the NAMED kernels only (demangled names without return type and
    parameter list, e.g. 'smx::k_pm_walk_len<2>'), in sorted order: stays the same when OTHER kernels are added to the library or change.
    -> 16 hex digits, or None when a tool, the library or one of the kernels is missing"""
    lib = lib or os.path.join(ROOT, "spades_amd", "csrc", "libspades_mi355x.so")
    try:
        with tempfile.TemporaryDirectory() as td:
            co = _code_object(lib, td)
            secs = {}
            for line in subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-S", "--wide", co], text=True).splitlines():
                f = line.replace("[", " ").replace("]", " ").split()
                if len(f) >= 6 and f[1] in (".text", ".rodata"):
                    secs[f[1]] = (int(f[3], 16), int(f[4], 16))  # address, file offset
            syms = []
            for line in subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], text=True).splitlines():
                f = line.split()
                if len(f) >= 8 and f[3] in ("FUNC", "OBJECT") and f[0].rstrip(":").isdigit():
                    syms.append((f[7], int(f[1], 16), int(f[2]), f[3]))
            dem = subprocess.check_output(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n, *_ in syms), text=True).splitlines()
            data = open(co, "rb").read()
            table = {}
            for (mangled, addr, size, typ), d in zip(syms, dem):
                base = d[5:] if d.startswith("void ") else d
                base = base.split("(")[0]
                sec = ".rodata" if mangled.endswith(".kd") else ".text"
                if sec not in secs or (typ == "OBJECT") != mangled.endswith(".kd"):
                    continue
                off = addr - secs[sec][0] + secs[sec][1]
                table.setdefault(base, {})["kd" if sec == ".rodata" else "code"] = data[off:off + size]
            h = hashlib.sha256()
            for n in sorted(set(names)):
                e = table.get(n)
                if not e or "code" not in e or "kd" not in e or not e["code"]:
                    return None
                kd = e["kd"]
                if len(kd) >= 24:  # bytes 16..23 of a descriptor: the offset from the descriptor to the code — a matter of layout, not of the kernel
                    kd = kd[:16] + kd[24:]
                h.update(n.encode() + b"\0" + e["code"] + kd)
            return h.hexdigest()[:16]
    except (OSError, subprocess.CalledProcessError, ValueError, IndexError):
        return None


def pmc_table_kernels(csv_path):
    """the kernel names of a PMC traffic table (tools/pmc_summary.py)"""
    out = []
    for line in open(csv_path):
        f = line.strip().rsplit(",", 5)
        if len(f) >= 6 and f[0] not in ("kernel", "TOTAL") and not f[0].startswith("#"):
            out.append(f[0])
    return out


if __name__ == "__main__":
    lib_ = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None
    print("all kernels:", device_code_hash(lib_))
    csvs = [a for a in sys.argv[1:] if a.endswith(".csv")]
    for c in csvs:
        print(c, "kernels of the table:", kernel_code_hash(pmc_table_kernels(c), lib_))
