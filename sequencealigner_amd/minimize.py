"""Host-side mirror of the reference's minimizer interface over the C-ABI (include/sa_hip.h).

`MinimizeBinary()` / `Minimize()` take (sequence, sequence_len, kmer_len, window_len) like the reference
functions (src/minimizer_binary.h:1-4, src/minimizer.h:2-5) and return the sorted, de-duplicated list of
(value, position, strand) tuples.  `DeviceMinimizeBatch` keeps a batch of sequences resident in HBM;
`run()` only enqueues work.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _as_u8(x):
    if isinstance(x, str):
        x = x.encode()
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


class DeviceMinimizeBatch:
    def __init__(self, seqs, k, w, variant=capi.SA_MIN_BINARY, device=0, out_capacity=None):
        from . import synth
        blob, off = synth.to_blob(seqs if isinstance(seqs, np.ndarray) and seqs.ndim == 2
                                  else [_as_u8(s) for s in seqs])
        lens = np.diff(off.astype(np.int64))
        self.lens = lens
        dev = torch.device("cuda", device)
        # the kernels fetch bases with 8-byte loads: keep 8 bytes of slack behind the blob
        d_seq = torch.from_numpy(np.concatenate([blob, np.zeros(16, np.uint8)])).to(dev)
        d_off = torch.from_numpy(off.astype(np.uint64).view(np.int64)).to(dev)
        windows = int((lens - (int(k) + int(w) - 2)).clip(min=0).sum())
        self._setup(d_seq, d_off, len(off) - 1, int(off[-1]), int(lens.min()), int(lens.max()), k, w, variant,
                    device, out_capacity, windows)

    @classmethod
    def from_device(cls, d_seq, d_off, n, total_len, min_len, max_len, k, w, variant=capi.SA_MIN_BINARY,
                    device=0, out_capacity=None):
        """A batch over sequences that already lie in device memory (d_seq: uint8 blob with 8 bytes of slack
        behind total_len, d_off: int64 [n + 1]), e.g. as ingest.DeviceRecords holds them."""
        self = cls.__new__(cls)
        self.lens = None
        self._setup(d_seq, d_off, n, total_len, min_len, max_len, k, w, variant, device, out_capacity, None)
        return self

    def _setup(self, d_seq, d_off, n, total_len, min_len, max_len, k, w, variant, device, out_capacity, windows):
        self.lib = capi.lib()
        self.device = torch.device("cuda", device)
        self.dev_index = device
        self.n = int(n)
        self.k, self.w, self.variant = int(k), int(w), int(variant)
        self.total_len = int(total_len)
        self.max_len, self.min_len = int(max_len), int(min_len)
        self.windows = windows
        self.d_seq, self.d_off = d_seq, d_off
        cap = self.lib.sa_minimize_out_capacity(self.n, self.total_len, self.k, self.w)
        self.out_capacity = int(out_capacity if out_capacity is not None else cap)
        self.d_out = torch.empty(max(self.out_capacity, 1) * 3, dtype=torch.int32, device=self.device)
        self.d_out_off = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
        self.d_status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.workspace_bytes = self.lib.sa_minimize_workspace_bytes(self.n, self.total_len, self.max_len,
                                                                    self.k, self.w, self.variant)
        self.d_ws = torch.empty(max(self.workspace_bytes, 1), dtype=torch.uint8, device=self.device)
        self.batch = capi.MinimizeBatch(
            self.d_seq.data_ptr(), self.d_off.data_ptr(), self.n, self.total_len, self.max_len,
            self.min_len, self.k, self.w, self.variant, self.d_out.data_ptr(), self.out_capacity,
            self.d_out_off.data_ptr(), self.d_status.data_ptr())

    def run(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        capi.check(self.lib.sa_minimize_batch_device(self.dev_index, C.c_void_p(s.cuda_stream),
                                                     C.byref(self.batch),
                                                     C.c_void_p(self.d_ws.data_ptr()),
                                                     self.workspace_bytes))

    def status(self):
        return int(self.d_status.item())

    def results(self):
        """-> list of (m, 3) uint32 arrays (value, position, strand), one per sequence."""
        st = self.status()
        if st != capi.SA_OK:
            raise capi.SaError(st, "device reported %s" % {capi.SA_ERR_RANGE: "k-mer value beyond INT_MAX",
                                                          capi.SA_ERR_CAPACITY: "output capacity exceeded",
                                                          capi.SA_ERR_ARGUMENT: "decimal Minimize: a byte other "
                                                                                "than ACGT / acgt"}.get(st, st))
        off = self.d_out_off.cpu().numpy()
        out = self.d_out[: int(off[-1]) * 3].cpu().numpy().view(np.uint32).reshape(-1, 3)
        return [out[off[i]:off[i + 1]] for i in range(self.n)]

    def counts(self):
        return np.diff(self.d_out_off.cpu().numpy())


def minimize_binary(seq, k, w, device=0):
    """(m, 3) uint32 array of the minimizers of one sequence (MinimizeBinary semantics)."""
    b = DeviceMinimizeBatch([seq], k, w, capi.SA_MIN_BINARY, device)
    b.run()
    torch.cuda.synchronize(b.device)
    return b.results()[0]


def minimize_batch(seqs, k, w, variant=capi.SA_MIN_BINARY, device=0):
    b = DeviceMinimizeBatch(seqs, k, w, variant, device)
    b.run()
    torch.cuda.synchronize(b.device)
    return b.results()


def _tuples(arr):
    return [(int(v), int(p), bool(s)) for v, p, s in arr]


def MinimizeBinary(sequence, sequence_len, kmer_len, window_len, device=0):
    """Mirror of MinimizeBinary (src/minimizer_binary.cpp:136-178): list of (value, position, strand)."""
    return _tuples(minimize_binary(_as_u8(sequence)[:sequence_len], kmer_len, window_len, device))


def Minimize(sequence, sequence_len, kmer_len, window_len, device=0):
    """Mirror of Minimize (src/minimizer.cpp:124-214).  Raises OverflowError where the reference's stoi
    throws std::out_of_range (a k-mer's decimal value above INT_MAX)."""
    try:
        res = minimize_batch([_as_u8(sequence)[:sequence_len]], kmer_len, window_len,
                             capi.SA_MIN_DECIMAL, device)
    except capi.SaError as e:
        if e.code == capi.SA_ERR_RANGE:
            raise OverflowError("stoi") from e
        raise
    return _tuples(res[0])
