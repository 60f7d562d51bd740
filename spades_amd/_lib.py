"""ctypes binding of libspades_mi355x.so (include/smx.h). Fails loudly when the library is missing."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libspades_mi355x.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "smx.h")

OK = 0
INVALID_INPUT_FORMAT, INPUT_FILE_NOT_FOUND, IO_ERROR, INVALID_PARAMETER, MEMORY_LIMIT_EXCEEDED, DEVICE_ERROR = 64, 65, 66, 67, 68, 70
MODE_ALL, MODE_CANONICAL = 0, 1

_lib = None


def declared_symbols():
    """Every function include/smx.h declares (used by the CPU-side export test)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). spades_amd has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's). Whichever copy is mapped
    # first serves both; with the system copy first, torch later reports "No HIP GPUs are available" (measured). The Python host
    # side shares the process with torch (device memory, streams, torch.distributed), so torch's copy goes first when torch exists.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    sig = {
        "smx_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_size_t]),
        "smx_destroy": (None, [vp]),
        "smx_trim": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
        "smx_arena_free_bytes": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
        "smx_prewarm": (C.c_int, [vp, C.c_size_t, C.c_size_t]),
        "smx_pool_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "smx_pool_free": (C.c_int, [vp, vp]),
        "smx_graph_clear": (C.c_int, [vp]),
        "smx_last_error": (C.c_char_p, [vp]),
        "smx_version": (C.c_char_p, []),
        "smx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
        "smx_reads_clear": (C.c_int, [vp]),
        "smx_submit_reads_ascii": (C.c_int, [vp, C.c_char_p, u64p, C.c_uint64]),
        "smx_submit_reads_packed": (C.c_int, [vp, u64p, C.c_uint64, u64p, u32p, C.c_uint64]),
        "smx_submit_reads_device": (C.c_int, [vp, vp, C.c_uint64, vp, vp, C.c_uint64]),
        "smx_graph_tip_stats": (C.c_int, [vp, u64p]),
        "smx_graph_route_stats": (C.c_int, [vp, u64p]),
        "smx_graph_copy_flanking": (C.c_int, [vp, u32p, u32p]),
        "smx_build_graph_from_records": (C.c_int, [vp, C.c_uint, C.c_uint, vp, C.c_uint64]),
        "smx_graph_set_coverage": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_uint64]),
        "smx_graph_shard_updates": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, vp, C.c_uint64, u64p]),
        "smx_graph_shard_build": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, C.c_uint64]),
        "smx_graph_shard_info": (C.c_int, [vp, u64p, u64p]),
        "smx_graph_shard_copy": (C.c_int, [vp, vp, vp]),
        "smx_build_graph_from_kmers": (C.c_int, [vp, C.c_uint, C.c_uint, vp, vp, C.c_uint64, u64p, C.c_uint64]),
        "smx_graph_set_kpomers": (C.c_int, [vp, vp, C.c_uint64, u64p]),
        "smx_copy_kmers_device": (C.c_int, [vp, vp]),
        "smx_copy_bucket_device": (C.c_int, [vp, C.c_uint, vp]),
        "smx_submit_fastq_text": (C.c_int, [vp, C.c_char_p, C.c_uint64, C.c_int, u64p, u64p]),
        "smx_pinned_alloc": (vp, [C.c_size_t]),
        "smx_pinned_free": (None, [vp]),
        "smx_submit_reads_binary": (C.c_int, [vp, C.c_char_p]),
        "smx_reads_info": (C.c_int, [vp, u64p, u64p]),
        "smx_count": (C.c_int, [vp, C.c_uint, C.c_int, C.c_uint]),
        "smx_count_info": (C.c_int, [vp, u64p, C.POINTER(C.c_uint), u64p]),
        "smx_bucket_sizes": (C.c_int, [vp, u64p]),
        "smx_copy_bucket": (C.c_int, [vp, C.c_uint, vp]),
        "smx_copy_final_kmers": (C.c_int, [vp, vp]),
        "smx_write_final_kmers": (C.c_int, [vp, C.c_char_p]),
        "smx_count_to_file": (C.c_int, [vp, C.c_uint, C.c_int, C.c_uint, C.c_char_p]),
        "smx_device_kmers": (vp, [vp]),
        "smx_extract_count": (C.c_int, [vp, C.c_uint, C.c_int, u64p]),
        "smx_extract_partition": (C.c_int, [vp, C.c_uint, C.c_int, C.c_uint, C.c_uint, vp, C.c_uint64, u64p]),
        "smx_extract_partition_owned": (C.c_int, [vp, C.c_uint, C.c_int, C.c_uint, C.c_uint, C.POINTER(vp), u64p]),
        "smx_extract_release": (C.c_int, [vp]),
        "smx_exchange_buffer": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
        "smx_exchange_release": (C.c_int, [vp]),
        "smx_kmers_with_masks_supported": (C.c_int, [C.c_uint]),
        "smx_extract_kmers_ext_owned": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(vp), u64p]),
        "smx_graph_shard_from_ext": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, C.c_uint64]),
        "smx_graph_shard_ext_stats": (C.c_int, [vp, u64p]),
        "smx_shard_walk_counts": (C.c_int, [vp, u64p, u64p]),
        "smx_shard_walk_requests": (C.c_int, [vp, C.c_int, C.c_uint, vp, u64p, u64p]),
        "smx_shard_walk_requests_range": (C.c_int, [vp, C.c_int, C.c_uint, C.c_uint64, C.c_uint64, vp, u64p, u64p]),
        "smx_shard_walk_starts": (C.c_int, [vp, u64p]),
        "smx_shard_walks": (C.c_int, [vp, u64p, vp, u64p]),
        "smx_shard_walk_loops": (C.c_int, [vp, u64p]),
        "smx_shard_lookup": (C.c_int, [vp, vp, C.c_uint64, u64p]),
        "smx_shard_gather_kmers": (C.c_int, [vp, u64p, C.c_uint64, vp, C.POINTER(C.c_uint8)]),
        "smx_shard_unitigs": (C.c_int, [vp, C.c_uint64, u64p, u64p, u64p, C.POINTER(C.c_uint8), u64p, u64p]),
        "smx_shard_unitigs_copy": (C.c_int, [vp, u64p, u64p, u64p, u64p, C.POINTER(C.c_uint8)]),
        "smx_build_graph_from_unitigs": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, u64p, C.c_uint64, u64p, u64p, u64p, C.POINTER(C.c_uint8),
                                                  C.c_uint64, u64p, u64p, C.POINTER(C.c_uint8), C.c_uint64]),
        "smx_graph_fingerprint": (C.c_int, [vp, u64p]),
        "smx_graph_fingerprint_portable": (C.c_int, [vp, u64p]),
        "smx_count_records": (C.c_int, [vp, C.c_uint, C.c_uint, vp, C.c_uint64]),
        "smx_rank_first_bucket": (C.c_uint, [C.c_uint, C.c_uint, C.c_uint]),
        "smx_last_timings": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
        "smx_build_graph": (C.c_int, [vp, C.c_uint, C.c_uint]),
        "smx_graph_info": (C.c_int, [vp, u64p]),
        "smx_graph_copy_kmers": (C.c_int, [vp, vp, vp]),
        "smx_graph_copy_unitigs": (C.c_int, [vp, u64p, vp]),
        "smx_graph_write_gfa": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
        "smx_graph_fill_coverage": (C.c_int, [vp]),
        "smx_graph_copy_coverage": (C.c_int, [vp, u32p]),
        "smx_graph_coverage_histogram": (C.c_int, [vp, u64p, C.c_uint64, u64p]),
        "smx_graph_write_unitigs": (C.c_int, [vp, C.c_char_p]),
        "smx_graph_write_spades": (C.c_int, [vp, C.c_char_p]),
        "smx_graph_write_fastg": (C.c_int, [vp, C.c_char_p]),
        "smx_host_write_graph": (C.c_int, [C.c_uint, C.c_uint64, u64p, C.c_char_p, u64p, u64p, u32p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
