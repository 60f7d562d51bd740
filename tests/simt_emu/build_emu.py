#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds tests/simt_emu/_build/libspades_emu.so — the library's single translation unit (spades_amd/csrc/smx_api.hip
with everything it includes: C ABI, host pipeline, every gfx950 kernel as written) compiled by g++ against tests/simt_emu/hip/hip_runtime.h,
the fiber-based SIMT stand-in. The sources are used as they are except for what only an AMDGPU assembler understands:
  * inline `asm volatile("s_waitcnt ...")` (waits that order nothing in a sequential emulation) and the library's LDS barrier (s_barrier),
  * `extern __shared__ T name[];` (dynamic LDS: a pointer to the emulator's one LDS buffer),
  * a few clang builtins / vector attributes g++ lacks, and the register class of empty optimizer-barrier asm statements ("+v" -> "+r").
The transformed copies live under _build/src; nothing of this is ever loaded by spades_amd (the product has no CPU path)."""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "spades_amd", "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libspades_emu.so")


def transform(text):
    text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)\\n\\ts_barrier" ::: "memory"\);', "EMU_LDS_BARRIER();", text)
    text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', "/* s_waitcnt: nothing to wait for here */;", text)
    text = text.replace("__builtin_amdgcn_wave_barrier();", "EMU_WAVE_BARRIER();")
    # an EMPTY asm statement that only hides a value from the optimizer ("+v": a vector register on gfx950) — a general register here
    text = re.sub(r'asm volatile\(""\s*:\s*"\+v"\((\w+)\)\);', r'__asm__ __volatile__("" : "+r"(\1));', text)
    text = re.sub(r"extern __shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_0-9 ]+?)\s+(\w+)\[\];", r"\1 *\2 = (\1 *)emu::g_ctx->lds;", text)
    text = text.replace("__builtin_rotateleft32", "emu_rotl32")
    # clang's vector extension and its cache-policy builtins (non-temporal = a hint to the memory system: a plain access here)
    text = text.replace("__attribute__((ext_vector_type(2)))", "__attribute__((vector_size(16)))")
    text = re.sub(r"__builtin_nontemporal_store\(", "emu_nt_store(", text)
    text = re.sub(r"__builtin_nontemporal_load\(", "emu_nt_load(", text)
    assert "asm volatile" not in text and "extern __shared__" not in text, "an AMDGPU-only construct the transform does not know"
    return text


FLAGS_STAMP = os.path.join(BUILD, "cxxflags.txt")  # what EMU_CXXFLAGS the library was built with (a sanitizer build must not be taken for the plain one)


def stale():
    if not os.path.exists(LIB):
        return True
    if (open(FLAGS_STAMP).read() if os.path.exists(FLAGS_STAMP) else "") != os.environ.get("EMU_CXXFLAGS", ""):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + [os.path.join(ROOT, "include", "smx.h"),
                                                                                                os.path.join(HERE, "hip", "hip_runtime.h"), __file__]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False):
    if not force and not stale():
        return LIB
    src = os.path.join(BUILD, "src", "spades_amd", "csrc")
    shutil.rmtree(os.path.join(BUILD, "src"), ignore_errors=True)
    os.makedirs(src)
    os.makedirs(os.path.join(BUILD, "src", "include"))
    shutil.copy(os.path.join(ROOT, "include", "smx.h"), os.path.join(BUILD, "src", "include", "smx.h"))
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp")):
            open(os.path.join(src, f), "w").write(transform(open(os.path.join(CSRC, f)).read()))
    cmd = ["g++", "-std=c++17", "-O1", "-g0", "-fPIC", "-shared", "-pthread", "-x", "c++", "-DEMU_DEFINE_SWITCH", "-Wno-unknown-pragmas", "-Wno-attributes",
           "-fno-omit-frame-pointer", *os.environ.get("EMU_CXXFLAGS", "").split(), "-I", HERE, "-include", os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(src, "smx_api.hip"), "-o", LIB]
    subprocess.check_call(cmd)
    open(FLAGS_STAMP, "w").write(os.environ.get("EMU_CXXFLAGS", ""))
    return LIB


def build_tools():
    """the two CLI clones (spades_amd/tools/*_main.cpp as they are) linked against the emulated library: tests/simt_emu/_build/spades-*-mi355x.
    (Their --gpus N hosts call hip / RCCL directly: those paths stay for the GPU box; libamdhip64 / librccl are only link-time names here.)"""
    lib = build()
    tools = os.path.join(ROOT, "spades_amd", "tools")
    out = []
    for t in ("kmercount", "gbuilder"):
        exe = os.path.join(BUILD, f"spades-{t}-mi355x")
        srcs = [os.path.join(tools, f) for f in os.listdir(tools) if f.endswith((".cpp", ".hpp"))]
        if not os.path.exists(exe) or any(os.path.getmtime(x) > os.path.getmtime(exe) for x in srcs + [lib]):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(tools, f"{t}_main.cpp"), "-L" + BUILD, "-lspades_emu", "-lz", "-pthread",
                                   "-Wl,-rpath," + BUILD, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
                                   "-Wl,-rpath,/opt/rocm/lib"])
        out.append(exe)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
