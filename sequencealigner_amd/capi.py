"""ctypes binding of the C-ABI in include/sa_hip.h (csrc/libsa_hip.so).

There is no CPU fallback: if the library is missing or cannot be loaded the import of the compute
path fails with an error saying how to build it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsa_hip.so")

SA_GLOBAL, SA_LOCAL, SA_SEMIGLOBAL = 0, 1, 2
SA_MIN_BINARY, SA_MIN_DECIMAL = 0, 1
(SA_OK, SA_ERR_HIP, SA_ERR_ARGUMENT, SA_ERR_WORKSPACE, SA_ERR_NO_DEVICE, SA_ERR_RANGE, SA_ERR_CAPACITY,
 SA_ERR_FORMAT, SA_ERR_MEMORY) = range(9)
SA_FMT_FASTA, SA_FMT_FASTQ = 0, 1
SA_UNKNOWN_TYPE_SCORE = 9999


class SaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sa_hip error {code}: {msg}")
        self.code = code


class AlignBatch(C.Structure):
    _fields_ = [
        ("d_q", C.c_void_p), ("d_q_off", C.c_void_p), ("d_t", C.c_void_p), ("d_t_off", C.c_void_p),
        ("n", C.c_uint32), ("max_qlen", C.c_uint32), ("max_tlen", C.c_uint32),
        ("min_qlen", C.c_uint32), ("min_tlen", C.c_uint32),
        ("type", C.c_int), ("match", C.c_int), ("mismatch", C.c_int), ("gap", C.c_int),
        ("d_score", C.c_void_p), ("d_target_begin", C.c_void_p), ("d_cigar_runs", C.c_void_p),
        ("d_cigar_nruns", C.c_void_p), ("cigar_stride", C.c_uint64),
    ]


class MinimizeBatch(C.Structure):
    _fields_ = [
        ("d_seq", C.c_void_p), ("d_off", C.c_void_p), ("n", C.c_uint32), ("total_len", C.c_uint64),
        ("max_len", C.c_uint32), ("min_len", C.c_uint32), ("k", C.c_uint32), ("w", C.c_uint32),
        ("variant", C.c_int), ("d_out", C.c_void_p), ("out_capacity", C.c_uint64),
        ("d_out_off", C.c_void_p), ("d_status", C.c_void_p),
    ]


# every symbol include/sa_hip.h declares: (name, restype, argtypes)
class ChainBatch(C.Structure):
    _fields_ = [("d_ref", C.c_void_p), ("n_ref", C.c_uint32), ("frequent_fraction", C.c_float),
                ("d_reads", C.c_void_p), ("d_read_off", C.c_void_p), ("n_reads", C.c_uint32),
                ("n_read_triples", C.c_uint64), ("d_out", C.c_void_p), ("d_n_matches", C.c_void_p)]


class IngestBatch(C.Structure):
    _fields_ = [("d_file", C.c_void_p), ("n_bytes", C.c_uint64), ("format", C.c_int), ("d_seq", C.c_void_p),
                ("seq_capacity", C.c_uint64), ("d_seq_off", C.c_void_p), ("d_name_start", C.c_void_p),
                ("d_name_len", C.c_void_p), ("d_qual_start", C.c_void_p), ("max_records", C.c_uint64),
                ("n_records", C.c_uint64), ("n_bases", C.c_uint64), ("min_len", C.c_uint32),
                ("max_len", C.c_uint32)]


SYMBOLS = [
    ("sa_last_error", C.c_char_p, []),
    ("sa_device_count", C.c_int, []),
    ("sa_device_arch", C.c_char_p, [C.c_int]),
    ("sa_set_option", C.c_int, [C.c_char_p, C.c_long]),
    ("sa_align_workspace_bytes", C.c_size_t, [C.POINTER(AlignBatch), C.POINTER(C.c_size_t)]),
    ("sa_align_batch_device", C.c_int,
     [C.c_int, C.c_void_p, C.POINTER(AlignBatch), C.c_void_p, C.c_size_t]),
    ("sa_align_static_scratch_bytes", C.c_size_t, []),
    ("sa_align_batch_device_static", C.c_int, [C.c_int, C.c_void_p, C.POINTER(AlignBatch)]),
    ("sa_align_static_init", C.c_int, [C.c_int]),
    ("sa_align_may_synchronize", C.c_int, [C.c_int, C.POINTER(AlignBatch), C.c_size_t]),
    ("sa_align_kernel_name", C.c_char_p, [C.POINTER(AlignBatch)]),
    ("sa_align_profile", C.c_int, [C.c_int]),
    ("sa_align_profile_read", C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("sa_cigar_stats_device", C.c_int,
     [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("sa_align_batch_host", C.c_int,
     [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
      C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
      C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    ("sa_free", None, [C.c_void_p]),
    ("sa_minimize_workspace_bytes", C.c_size_t,
     [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    ("sa_minimize_out_capacity", C.c_uint64, [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]),
    ("sa_minimize_batch_device", C.c_int,
     [C.c_int, C.c_void_p, C.POINTER(MinimizeBatch), C.c_void_p, C.c_size_t]),
    ("sa_minimize_batch_host", C.c_int,
     [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_size_t, C.c_uint32, C.c_uint32, C.c_int,
      C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("sa_chain_batch_device", C.c_int,
     [C.c_int, C.c_void_p, C.POINTER(ChainBatch), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("sa_ingest_workspace_bytes", C.c_size_t, [C.c_uint64]),
    ("sa_ingest_device", C.c_int, [C.c_int, C.c_void_p, C.POINTER(IngestBatch), C.c_void_p, C.c_size_t]),
    ("sa_pack2bit_device", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("sa_unpack2bit_device", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("sa_gather_spans_device", C.c_int,
     [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("sa_shard_units", C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("sa_remove_duplicate_host", C.c_int,
     [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]),
]

_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        # torch ships its own HIP runtime; loading it first makes the process use one runtime whichever
        # of the two libraries is touched first (two runtimes in one process do not see each other's GPU)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def set_option(name, value):
    """sa_set_option(): tuning / test options of the process (include/sa_hip.h lists them)."""
    check(lib().sa_set_option(name.encode(), int(value)))


def check(code):
    if code != SA_OK:
        raise SaError(code, lib().sa_last_error().decode(errors="replace"))
