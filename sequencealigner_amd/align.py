"""Host-side mirror of the reference's alignment interface over the C-ABI (include/sa_hip.h).

`Align()` has the argument meaning of the reference's Align() (src/alignment.h:28-36):
query/target with explicit lengths, an AlignmentType, match/mismatch/gap, and the two optional
outputs cigar and target_begin.  `DeviceAlignBatch` is the resident-batch form used for throughput
work: the inputs live in HBM, `run()` only enqueues kernels.

torch is used for device memory and streams only.
"""
import ctypes as C
import enum

import numpy as np
import torch

from . import capi

_OPS = "MXID"


class AlignmentType(enum.IntEnum):
    """Same values as the reference's plain enum AlignmentType (src/alignment.h:25)."""
    Global = 0
    Local = 1
    Semiglobal = 2


def _as_u8(x):
    if isinstance(x, str):
        x = x.encode()
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def runs_to_cigar(runs):
    """uint32 runs ((len << 2) | op) -> "<count><op>..." as CompressCigar prints it
    (src/alignment.cpp:17-33: the count is always printed, including 1)."""
    return "".join(f"{int(v) >> 2}{_OPS[int(v) & 3]}" for v in runs)


class DeviceAlignBatch:
    """A batch of independent pairs resident on one GPU.

    queries/targets: either (n, L) uint8 arrays (uniform length) or lists of byte strings/arrays.
    """

    def __init__(self, queries, targets, type_, match, mismatch, gap, want_cigar=True,
                 want_target_begin=True, device=0, workspace_bytes=None):
        from . import synth
        qb, qo = synth.to_blob(queries if isinstance(queries, np.ndarray) else [_as_u8(x) for x in queries])
        tb, to = synth.to_blob(targets if isinstance(targets, np.ndarray) else [_as_u8(x) for x in targets])
        assert len(to) == len(qo)
        qlens = np.diff(qo.astype(np.int64))
        tlens = np.diff(to.astype(np.int64))
        ddev = torch.device("cuda", device)

        def dev(a, dtype):
            t = torch.from_numpy(np.ascontiguousarray(a).view(dtype) if a.size else np.zeros(1, dtype))
            return t.to(ddev)

        # +8 bytes of slack so that kernels may read whole dwords at the end of the blob
        self._setup(dev(np.concatenate([qb, np.zeros(8, np.uint8)]), np.uint8),
                    dev(qo.astype(np.uint64).view(np.int64), np.int64),
                    dev(np.concatenate([tb, np.zeros(8, np.uint8)]), np.uint8),
                    dev(to.astype(np.uint64).view(np.int64), np.int64), qlens, tlens, type_, match, mismatch, gap,
                    want_cigar, want_target_begin, device, workspace_bytes)

    @classmethod
    def from_device(cls, d_q, d_qo, d_t, d_to, qlens, tlens, type_, match, mismatch, gap, want_cigar=True,
                    want_target_begin=True, device=0, workspace_bytes=None):
        """A batch over pairs that already lie in device memory: uint8 blobs with 8 bytes of slack, int64
        offsets [n + 1]; qlens / tlens are the host copies of the lengths."""
        self = cls.__new__(cls)
        self._setup(d_q, d_qo, d_t, d_to, np.asarray(qlens, dtype=np.int64), np.asarray(tlens, dtype=np.int64),
                    type_, match, mismatch, gap, want_cigar, want_target_begin, device, workspace_bytes)
        return self

    def _setup(self, d_q, d_qo, d_t, d_to, qlens, tlens, type_, match, mismatch, gap, want_cigar,
               want_target_begin, device, workspace_bytes):
        self.lib = capi.lib()
        self.device = torch.device("cuda", device)
        self.dev_index = device
        self.n = len(qlens)
        self.qlens, self.tlens = qlens, tlens
        self.max_q = int(qlens.max()) if self.n else 0
        self.max_t = int(tlens.max()) if self.n else 0
        self.cells = int((qlens * tlens).sum())
        self.type, self.match, self.mismatch, self.gap = int(type_), int(match), int(mismatch), int(gap)
        self.d_q, self.d_qo, self.d_t, self.d_to = d_q, d_qo, d_t, d_to
        n1 = max(self.n, 1)
        self.d_score = torch.zeros(n1, dtype=torch.int32, device=self.device)
        self.d_tb = torch.zeros(n1, dtype=torch.int32, device=self.device) if want_target_begin else None
        self.stride = self.max_q + self.max_t
        if want_cigar:
            self.d_runs = torch.zeros(max(n1 * self.stride, 1), dtype=torch.int32, device=self.device)
            self.d_nruns = torch.zeros(n1, dtype=torch.int32, device=self.device)
        else:
            self.d_runs = self.d_nruns = None
        self.min_q = int(qlens.min()) if self.n else 0
        self.min_t = int(tlens.min()) if self.n else 0
        self.batch = capi.AlignBatch(
            self.d_q.data_ptr(), self.d_qo.data_ptr(), self.d_t.data_ptr(), self.d_to.data_ptr(),
            self.n, self.max_q, self.max_t, self.min_q, self.min_t, self.type, self.match,
            self.mismatch, self.gap,
            self.d_score.data_ptr(), self.d_tb.data_ptr() if self.d_tb is not None else None,
            self.d_runs.data_ptr() if self.d_runs is not None else None,
            self.d_nruns.data_ptr() if self.d_nruns is not None else None, self.stride)
        min_bytes = C.c_size_t(0)
        full = self.lib.sa_align_workspace_bytes(C.byref(self.batch), C.byref(min_bytes))
        if workspace_bytes is None:
            free, _ = torch.cuda.mem_get_info(self.device)
            workspace_bytes = min(full, max(min_bytes.value, int(free * 0.6)))
        self.workspace_bytes = max(int(workspace_bytes), min_bytes.value)
        self.d_ws = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)

    def kernel_name(self):
        return self.lib.sa_align_kernel_name(C.byref(self.batch)).decode()

    def run(self, stream=None):
        """Enqueue the whole batch on `stream` (a torch.cuda.Stream; default: the current stream)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        capi.check(self.lib.sa_align_batch_device(self.dev_index, C.c_void_p(s.cuda_stream),
                                                  C.byref(self.batch), C.c_void_p(self.d_ws.data_ptr()),
                                                  self.workspace_bytes))

    def run_static(self, stream=None):
        """The same batch through sa_align_batch_device_static: the library's own static scratch array takes
        the place of the workspace and nothing is allocated."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        capi.check(self.lib.sa_align_batch_device_static(self.dev_index, C.c_void_p(s.cuda_stream),
                                                         C.byref(self.batch)))

    def scores(self):
        return self.d_score[:self.n].cpu().numpy()

    def target_begins(self):
        return self.d_tb[:self.n].cpu().numpy().view(np.uint32)

    def cigar_stats(self, stream=None):
        """(alignment block length, per-op lengths [n, 4] in the order M X I D) of every pair, computed
        on the device from the run arrays - the sums the reference's mapper takes from the CIGAR text
        (getAlignmentBlockLengthAndMatchLength, src/mapper.cpp:371-399)."""
        import torch
        block = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
        ops = torch.zeros(max(self.n, 1) * 4, dtype=torch.int32, device=self.device)
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        capi.check(self.lib.sa_cigar_stats_device(self.device.index or 0, s.cuda_stream, self.n,
                                                  self.d_runs.data_ptr(), self.d_nruns.data_ptr(), self.stride,
                                                  block.data_ptr(), ops.data_ptr()))
        s.synchronize()
        return block[:self.n].cpu().numpy(), ops[:self.n * 4].cpu().numpy().reshape(self.n, 4)

    def cigars(self, indices=None):
        """CIGAR strings of the given pairs (default: all)."""
        nruns = self.d_nruns[:self.n].cpu().numpy()
        idx = range(self.n) if indices is None else indices
        out = []
        if indices is None:
            runs = self.d_runs.cpu().numpy().view(np.uint32)
            for p in idx:
                e = (p + 1) * self.stride
                out.append(runs_to_cigar(runs[e - nruns[p]:e]))
        else:
            for p in idx:
                e = (p + 1) * self.stride
                out.append(runs_to_cigar(self.d_runs[e - int(nruns[p]):e].cpu().numpy().view(np.uint32)))
        return out


def size_classes(qlens, tlens, few=64):
    """Groups pair indices by size class (powers of two of query and target length, at least 64): a device
    batch is planned for its longest query and target, so a few long outliers among many short pairs would
    make every pair pay for their matrix.  Classes of fewer than `few` pairs are put together into one batch,
    the last of the list: so few pairs run on kernels that give every pair its own wavefronts and its own
    number of steps, and are best started on a stream of their own beside the large classes.
    -> (list of index arrays, largest class first; True if the last one is that collection)."""
    qlens, tlens = np.asarray(qlens, dtype=np.int64), np.asarray(tlens, dtype=np.int64)
    key = (np.ceil(np.log2(np.maximum(tlens, 64))).astype(np.int64) << 8) | \
        np.ceil(np.log2(np.maximum(qlens, 64))).astype(np.int64)
    groups = [np.nonzero(key == k)[0] for k in sorted(set(key.tolist()), reverse=True)]
    large = [g for g in groups if g.size >= few]
    small = [g for g in groups if g.size < few]
    if small and large:
        return large + [np.sort(np.concatenate(small))], True
    return groups, False


def align_batch(queries, targets, type_, match, mismatch, gap, want_cigar=True, device=0):
    """-> (scores int32[n], cigars list[str] or None, target_begin uint32[n])"""
    b = DeviceAlignBatch(queries, targets, type_, match, mismatch, gap, want_cigar, True, device)
    b.run()
    torch.cuda.synchronize(b.device)
    return b.scores(), (b.cigars() if want_cigar else None), b.target_begins()


def Align(query, query_len, target, target_len, type_, match, mismatch, gap, want_cigar=True,
          want_target_begin=True, device=0):
    """Mirror of the reference Align() (src/alignment.cpp:353-374).

    Returns (score, cigar or None, target_begin or None).  An unknown type returns 9999 like the
    reference.  With want_cigar=False and want_target_begin=True, Global/Semiglobal report
    target_begin = 1 (the reference inspects an empty CIGAR string there).
    """
    if int(type_) not in (0, 1, 2):
        return capi.SA_UNKNOWN_TYPE_SCORE, None, None
    q = _as_u8(query)[:query_len]
    t = _as_u8(target)[:target_len]
    scores, cigars, tb = align_batch([q], [t], type_, match, mismatch, gap, want_cigar, device)
    tbv = None
    if want_target_begin:
        tbv = int(tb[0]) if (want_cigar or int(type_) == AlignmentType.Local) else 1
    return int(scores[0]), (cigars[0] if want_cigar else None), tbv
