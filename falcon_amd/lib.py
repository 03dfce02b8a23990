"""Loader for libfalcon_amd.so (HIP kernels + C ABI, include/falcon_amd.h).

The library is built in-tree by ``make -C falcon_amd/csrc`` (or
``__graft_entry__.build()``).  There is no CPU fallback anywhere in this
package: if the shared object is missing, or no HIP device is usable, the
product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (FALCON_AMD_LIB: another build of the same library -- kernel experiments)
SO_PATH = os.environ.get("FALCON_AMD_LIB") or os.path.join(HERE, "libfalcon_amd.so")


class FalconAmdError(RuntimeError):
    pass


class FaStats(C.Structure):
    _fields_ = [("L", C.c_longlong), ("C", C.c_longlong), ("D", C.c_longlong),
                ("A", C.c_longlong), ("T", C.c_longlong), ("O", C.c_longlong),
                ("n_piles", C.c_longlong), ("n_seqs", C.c_longlong),
                ("n_aligned", C.c_longlong),
                ("ms_index", C.c_float), ("ms_chain", C.c_float), ("ms_align", C.c_float),
                ("ms_consensus", C.c_float), ("ms_total", C.c_float),
                ("align_slots", C.c_int),
                ("ms_tags", C.c_float), ("ms_links", C.c_float), ("ms_score", C.c_float),
                ("ms_backtrace", C.c_float),
                ("align_slot_cells", C.c_longlong), ("align_relaunched", C.c_int),
                ("n_piles_failed", C.c_int),
                ("align_arena_bytes", C.c_longlong),
                ("align_pair_iterations", C.c_longlong), ("align_single_iterations", C.c_longlong),
                ("align_placements", C.c_longlong), ("align_parkings", C.c_longlong),
                ("align_handed_back", C.c_longlong), ("align_wide_rows", C.c_longlong),
                ("align_replacements", C.c_longlong),
                ("align_handed_back_tape", C.c_longlong), ("align_handed_back_wide", C.c_longlong),
                ("align_handed_back_escapes", C.c_longlong)]

    def b_alg(self) -> int:
        """Algorithmic bytes: L/4 + 4C + 8D + 16A + 12T + 5O."""
        return self.L // 4 + 4 * self.C + 8 * self.D + 16 * self.A + 12 * self.T + 5 * self.O


class Alignment(C.Structure):
    """src/c/common.h:59-69"""
    _fields_ = [("aln_str_size", C.c_int), ("dist", C.c_int), ("aln_q_s", C.c_int),
                ("aln_q_e", C.c_int), ("aln_t_s", C.c_int), ("aln_t_e", C.c_int),
                ("q_aln_str", C.c_void_p), ("t_aln_str", C.c_void_p)]


_lib = None


def load() -> C.CDLL:
    """dlopen the product library and declare the batch ABI prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FalconAmdError(
            "%s is missing: build it with `make -C falcon_amd/csrc` "
            "(there is no CPU fallback)" % SO_PATH)
    lib = C.CDLL(SO_PATH)
    lib.fa_last_error.restype = C.c_char_p
    lib.fa_device_count.restype = C.c_int
    lib.fa_warm.restype = C.c_int
    lib.fa_warm.argtypes = [C.c_void_p, C.c_longlong]
    lib.fa_create.restype = C.c_void_p
    lib.fa_create.argtypes = [C.c_int]
    lib.fa_destroy.argtypes = [C.c_void_p]
    lib.fa_batch_create.restype = C.c_void_p
    lib.fa_batch_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                    C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    lib.fa_batch_run.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_double]
    lib.fa_batch_submit.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_double]
    lib.fa_batch_wait.argtypes = [C.c_void_p]
    lib.fa_batch_fasta.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_longlong)]
    lib.fa_batch_fasta.restype = C.c_int
    lib.fa_fasta_records.argtypes = [C.c_char_p, C.c_char_p, C.c_longlong, C.c_int, C.c_char_p, C.c_longlong]
    lib.fa_fasta_records.restype = C.c_longlong
    lib.fa_batch_fetch.argtypes = [C.c_void_p, C.c_int]
    lib.fa_batch_trim_windows.argtypes = [C.c_void_p, C.c_uint, C.c_int]
    lib.fa_batch_result.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                    C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_int))]
    lib.fa_batch_stats.argtypes = [C.c_void_p, C.POINTER(FaStats)]
    lib.fa_batch_pile_error.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    lib.fa_batch_free.argtypes = [C.c_void_p]
    lib.fa_batch_range.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4 + \
        [C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.fa_batch_alignment.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 5 + \
        [C.POINTER(C.c_longlong)]
    lib.fa_debug_pack.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint), C.c_longlong]
    lib.fa_debug_pack.restype = C.c_int
    lib.fa_batch_debug_hits.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.fa_batch_debug_hits.restype = C.c_int
    lib.fa_batch_debug_tags.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_uint),
                                        C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_int)]
    lib.fa_batch_debug_tags.restype = C.c_int
    lib.fa_align_pairs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                   C.POINTER(C.c_int), C.c_int, C.c_int,
                                   C.POINTER(C.POINTER(Alignment))]
    lib.free_alignment.argtypes = [C.POINTER(Alignment)]
    # ingest (no GPU needed)
    lib.fa_reader_open.restype = C.c_void_p
    lib.fa_reader_open.argtypes = [C.c_int] * 6
    lib.fa_reader_next.restype = C.c_int
    lib.fa_reader_next.argtypes = [C.c_void_p, C.c_int, C.c_longlong,
                                   C.POINTER(C.POINTER(C.c_int)),
                                   C.POINTER(C.POINTER(C.c_char_p)),
                                   C.POINTER(C.POINTER(C.c_int)),
                                   C.POINTER(C.POINTER(C.c_char_p))]
    lib.fa_reader_keep.restype = C.c_int
    lib.fa_reader_keep.argtypes = [C.c_void_p, C.c_int]
    lib.fa_reader_error.restype = C.c_char_p
    lib.fa_reader_error.argtypes = [C.c_void_p]
    lib.fa_reader_close.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def last_error() -> str:
    return load().fa_last_error().decode("utf-8", "replace")
