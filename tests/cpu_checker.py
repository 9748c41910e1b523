"""ctypes binding of the CPU checkers (oracle/sa_oracle.h).  TEST INFRASTRUCTURE ONLY.

`restatement()` loads oracle/libsa_oracle.so (built from oracle/oracle.cpp); `reference()` loads
oracle/_ref/libsa_ref.so (the reference's own sources, compiled by oracle/Makefile) or returns None
where that build is absent.  Both expose the same interface.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class CpuChecker:
    def __init__(self, path):
        self.lib = L = C.CDLL(path)
        L.sa_cpu_align.restype = C.c_int
        L.sa_cpu_align.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        for f in (L.sa_cpu_minimize_binary, L.sa_cpu_minimize_decimal):
            f.restype = C.c_size_t
            f.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.sa_cpu_remove_duplicate.restype = C.c_size_t
        L.sa_cpu_remove_duplicate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.sa_cpu_align_batch.restype = C.c_int
        L.sa_cpu_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]
        L.sa_cpu_minimize_batch.restype = C.c_int
        L.sa_cpu_minimize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        if hasattr(L, "sa_cpu_score_linear"):
            L.sa_cpu_score_linear.restype = C.c_int
            L.sa_cpu_score_linear.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.c_int,
                                              C.c_int, C.c_int]
            L.sa_cpu_score_linear_batch.restype = C.c_int
            L.sa_cpu_score_linear_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        if hasattr(L, "sa_cpu_cigar_block_match"):
            L.sa_cpu_cigar_block_match.restype = None
            L.sa_cpu_cigar_block_match.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if hasattr(L, "sa_cpu_chain_batch"):
            L.sa_cpu_chain_batch.restype = None
            L.sa_cpu_chain_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_void_p]
        if hasattr(L, "sa_cpu_parse"):
            L.sa_cpu_parse.restype = C.c_int64
            L.sa_cpu_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_uint64]
        L.sa_cpu_kind.restype = C.c_char_p
        self.kind = L.sa_cpu_kind().decode()

    def align(self, query, target, type_, match, mismatch, gap, want_cigar=True,
              want_target_begin=True):
        """-> (score, cigar or None, target_begin or None)"""
        q = query if isinstance(query, bytes) else query.encode()
        t = target if isinstance(target, bytes) else target.encode()
        cap = 2 * (len(q) + len(t)) + 32
        buf = C.create_string_buffer(cap) if want_cigar else None
        clen = C.c_size_t(0)
        tb = C.c_uint32(0)
        score = self.lib.sa_cpu_align(q, len(q), t, len(t), type_, match, mismatch, gap, buf,
                                      cap if want_cigar else 0, C.byref(clen),
                                      C.byref(tb) if want_target_begin else None)
        return (score, buf.value.decode() if want_cigar else None,
                tb.value if want_target_begin else None)

    def score_linear(self, query, target, type_, match, mismatch, gap):
        """The score alone, in linear memory (restatement library only)."""
        q = query if isinstance(query, bytes) else query.encode()
        t = target if isinstance(target, bytes) else target.encode()
        return self.lib.sa_cpu_score_linear(q, len(q), t, len(t), type_, match, mismatch, gap)

    def score_linear_batch(self, q_blob, q_off, t_blob, t_off, type_, match, mismatch, gap, threads):
        n = len(q_off) - 1
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        scores = np.zeros(n, dtype=np.int32)
        self.lib.sa_cpu_score_linear_batch(q_blob.ctypes.data, q_off.ctypes.data, t_blob.ctypes.data,
                                           t_off.ctypes.data, n, type_, match, mismatch, gap, threads,
                                           scores.ctypes.data)
        return scores

    def _minimize(self, fn, seq, k, w):
        s = seq if isinstance(seq, bytes) else seq.encode()
        cap = max(len(s), 1)
        out = np.zeros((cap, 3), dtype=np.uint32)
        n = fn(s, len(s), k, w, out.ctypes.data, cap)
        if n == C.c_size_t(-1).value:
            return None
        return [tuple(int(x) for x in row) for row in out[:n]]

    def minimize_binary(self, seq, k, w):
        return self._minimize(self.lib.sa_cpu_minimize_binary, seq, k, w)

    def minimize_binary_array(self, seq, k, w):
        """Same as minimize_binary but returns an (n, 3) uint32 array (for long sequences)."""
        s = seq if isinstance(seq, bytes) else seq.encode()
        out = np.zeros((max(len(s), 1), 3), dtype=np.uint32)
        n = self.lib.sa_cpu_minimize_binary(s, len(s), k, w, out.ctypes.data, out.shape[0])
        return out[:n].copy()

    def minimize_decimal(self, seq, k, w):
        """None where the reference throws from stoi."""
        return self._minimize(self.lib.sa_cpu_minimize_decimal, seq, k, w)

    def remove_duplicate(self, triples):
        a = np.ascontiguousarray(np.asarray(triples, dtype=np.uint32).reshape(-1, 3))
        out = np.zeros_like(a)
        n = self.lib.sa_cpu_remove_duplicate(a.ctypes.data, a.shape[0], out.ctypes.data)
        return [tuple(int(x) for x in r) for r in a], [tuple(int(x) for x in r) for r in out[:n]]

    def chain_batch(self, ref_triples, read_triples, read_off, f):
        """(chain length, t_begin, t_end, q_begin, q_end) per read and the matched pairs per read, as the
        reference's mapper derives them from the minimizer lists."""
        ref = np.ascontiguousarray(ref_triples, dtype=np.uint32)
        rd = np.ascontiguousarray(read_triples, dtype=np.uint32)
        off = np.ascontiguousarray(read_off, dtype=np.uint64)
        n = len(off) - 1
        out = np.zeros((n, 5), dtype=np.int32)
        nm = np.zeros(n, dtype=np.uint64)
        self.lib.sa_cpu_chain_batch(ref.ctypes.data, len(ref), f, rd.ctypes.data, off.ctypes.data, n,
                                    out.ctypes.data, nm.ctypes.data)
        return out, nm

    def parse(self, data, fmt):
        """File bytes -> (seq blob, seq_off, name_start, name_len, qual_start) as numpy arrays, or None for a
        file that breaks the format (fmt 0 = FASTA, 1 = FASTQ)."""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        n = buf.size
        cap = n // 2 + 2
        seq = np.zeros(max(n, 1), dtype=np.uint8)
        off = np.zeros(cap + 1, dtype=np.uint64)
        ns = np.zeros(cap, dtype=np.uint64)
        nl = np.zeros(cap, dtype=np.uint32)
        qs = np.zeros(cap, dtype=np.uint64)
        r = self.lib.sa_cpu_parse(buf.ctypes.data if n else None, n, fmt, seq.ctypes.data, off.ctypes.data,
                                  ns.ctypes.data, nl.ctypes.data, qs.ctypes.data, cap)
        if r < 0:
            return None
        return seq[:int(off[r])].copy(), off[:r + 1].copy(), ns[:r].copy(), nl[:r].copy(), qs[:r].copy()

    def cigar_block_match(self, cigar):
        """(alignment block length, match length) as the reference's mapper computes them from CIGAR text."""
        abl, ml = C.c_int(0), C.c_int(0)
        self.lib.sa_cpu_cigar_block_match(cigar.encode(), C.byref(abl), C.byref(ml))
        return abl.value, ml.value

    def align_batch(self, q_blob, q_off, t_blob, t_off, type_, match, mismatch, gap, want_cigar,
                    threads):
        """Blobs are uint8 arrays, offsets uint64 arrays of n+1 entries. -> (scores, target_begin)"""
        n = len(q_off) - 1
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        scores = np.zeros(n, dtype=np.int32)
        tb = np.zeros(n, dtype=np.uint32)
        self.lib.sa_cpu_align_batch(q_blob.ctypes.data, q_off.ctypes.data, t_blob.ctypes.data,
                                    t_off.ctypes.data, n, type_, match, mismatch, gap,
                                    int(want_cigar), threads, scores.ctypes.data, tb.ctypes.data)
        return scores, tb

    def minimize_batch(self, blob, off, lens, k, w, threads):
        """-> (counts, hashes): per sequence the length of its minimizer list and the list's hash (the sum of
        sa_triple_hash over its entries, oracle/sa_oracle.h); `triple_list_hashes` computes the same figure
        from a device result."""
        n = len(lens)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        counts = np.zeros(n, dtype=np.uint64)
        hashes = np.zeros(n, dtype=np.uint64)
        self.lib.sa_cpu_minimize_batch(blob.ctypes.data, off.ctypes.data, lens.ctypes.data, n, k, w,
                                       threads, counts.ctypes.data, hashes.ctypes.data)
        return counts, hashes


def _i64(x):
    """The 64-bit pattern x as a signed Python int (torch has no unsigned 64-bit arithmetic)."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def triple_list_hashes(d_out, d_out_off):
    """Per-sequence hashes of minimizer lists that lie in device memory (d_out: int32 [3 * total] of
    (value, position, strand) triples, d_out_off: int64 [n + 1]), computed with torch on that device:
    the sum modulo 2^64 of sa_triple_hash (oracle/sa_oracle.h) over every list.  -> uint64 numpy array [n]."""
    import torch
    total = int(d_out_off[-1].item())
    n = d_out_off.numel() - 1
    tri = d_out[: 3 * total].reshape(-1, 3).to(torch.int64) & 0xFFFFFFFF
    first = d_out_off[:-1]
    counts = d_out_off[1:] - first
    seq = torch.repeat_interleave(torch.arange(n, device=d_out.device), counts)
    index = torch.arange(total, device=d_out.device) - first[seq]
    x = (tri[:, 0] | (tri[:, 1] << 32) | (tri[:, 2] << 63)) + (index + 1) * _i64(0x9E3779B97F4A7C15)
    x = x ^ ((x >> 30) & ((1 << 34) - 1))
    x = x * _i64(0xBF58476D1CE4E5B9)
    x = x ^ ((x >> 27) & ((1 << 37) - 1))
    x = x * _i64(0x94D049BB133111EB)
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    sums = torch.zeros(n, dtype=torch.int64, device=d_out.device)
    sums.index_add_(0, seq, x)
    return sums.cpu().numpy().view(np.uint64)


def _build(target):
    subprocess.run(["make", "-C", ORACLE_DIR, target], check=True, stdout=subprocess.DEVNULL)


_cache = {}


def restatement():
    if "r" not in _cache:
        path = os.path.join(ORACLE_DIR, "libsa_oracle.so")
        if not os.path.exists(path):
            _build("all")
        _cache["r"] = CpuChecker(path)
    return _cache["r"]


def reference():
    """The compiled reference, or None where oracle/_ref was not built (it cannot be built without
    the reference checkout, e.g. on the GPU box if the prebuilt file did not travel)."""
    if "ref" not in _cache:
        path = os.path.join(ORACLE_DIR, "_ref", "libsa_ref.so")
        _cache["ref"] = CpuChecker(path) if os.path.exists(path) else None
    return _cache["ref"]
