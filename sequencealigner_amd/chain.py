"""Seed chaining of reads against one reference on the GPU: the mapper's index / frequency filter / match /
longest-increasing-subsequence steps (reference src/mapper.cpp:172-352) over device-resident minimizer
lists.  No CPU fallback: the calls fail without the HIP library or a GPU."""
import ctypes as C

import numpy as np
import torch

from . import capi


def chain_batch(d_ref, d_reads, d_read_off, frequent_fraction=0.0, device=None, want_matches=False):
    """d_ref: int32/uint32 tensor [n_ref, 3] of the reference's (value, position, strand) triples;
    d_reads: [total, 3] triples of all reads; d_read_off: int64 tensor [n_reads + 1].
    Returns an int32 array [n_reads, 5] = (chain length, t_begin, t_end, q_begin, q_end) per read
    (and the matched pairs per read if want_matches)."""
    dev = d_ref.device if device is None else torch.device("cuda", device)
    n_reads = int(d_read_off.numel()) - 1
    total = int(d_read_off[-1].item())
    out = torch.zeros(max(n_reads, 1) * 5, dtype=torch.int32, device=dev)
    nm = torch.zeros(max(n_reads, 1), dtype=torch.int64, device=dev) if want_matches else None
    b = capi.ChainBatch()
    b.d_ref, b.n_ref, b.frequent_fraction = d_ref.data_ptr(), int(d_ref.shape[0]), float(frequent_fraction)
    b.d_reads, b.d_read_off, b.n_reads, b.n_read_triples = d_reads.data_ptr(), d_read_off.data_ptr(), n_reads, total
    b.d_out = out.data_ptr()
    b.d_n_matches = nm.data_ptr() if nm is not None else None
    lib = capi.lib()
    stream = torch.cuda.current_stream(dev)
    need = C.c_size_t(0)
    # a first guess that usually holds every match (the library says how much it needs when it does not)
    ws = torch.empty((4 << 20) + 64 * int(d_ref.shape[0]) + 48 * total, dtype=torch.uint8, device=dev)
    for _ in range(3):  # the number of matches sizes the workspace: grow it as the library asks
        rc = lib.sa_chain_batch_device(dev.index or 0, stream.cuda_stream, C.byref(b), ws.data_ptr(), ws.numel(),
                                       C.byref(need))
        if rc != capi.SA_ERR_WORKSPACE:
            break
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    capi.check(rc)
    stream.synchronize()
    res = out[:n_reads * 5].cpu().numpy().reshape(n_reads, 5)
    return (res, nm[:n_reads].cpu().numpy()) if want_matches else res
