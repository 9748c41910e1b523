"""FASTA / FASTQ ingest on the GPU: file bytes in device memory -> sequence blob + offsets for the batch
calls, plus name spans and length statistics.  Takes the place of the parser calls, the short / long
fragment split and PrintStats of the reference's mapper (reference src/mapper.cpp:532-577, :71-105).
No CPU fallback: the calls fail without the HIP library or a GPU."""
import ctypes as C

import numpy as np
import torch

from . import capi

FASTA, FASTQ = capi.SA_FMT_FASTA, capi.SA_FMT_FASTQ


def format_of(path):
    """The mapper's file-name tests, in its order (reference src/mapper.cpp:135-170, :540-561): any name
    containing ".fa", ".fasta" or ".fsa" is FASTA - which includes "reads.fastq" - then ".fastq",
    ".sanfastq", ".fq" are FASTQ; None for anything else (the mapper then parses nothing)."""
    if any(e in path for e in (".fa", ".fasta", ".fsa")):
        return FASTA
    if any(e in path for e in (".fastq", ".sanfastq", ".fq")):
        return FASTQ
    return None


class DeviceRecords:
    """The records of one file: d_seq (uint8 blob with 16 bytes of slack), d_seq_off (int64, n + 1),
    d_name_start / d_name_len / d_qual_start (spans of d_file), n, n_bases, min_len, max_len."""

    def __init__(self, d_file, fmt, device=0):
        dev = torch.device("cuda", device)
        self.device, self.format, self.d_file = dev, fmt, d_file
        n_bytes = int(d_file.numel())
        lib = capi.lib()
        stream = torch.cuda.current_stream(dev)
        ws = torch.empty(lib.sa_ingest_workspace_bytes(n_bytes), dtype=torch.uint8, device=dev)
        b = capi.IngestBatch()
        b.d_file, b.n_bytes, b.format = d_file.data_ptr(), n_bytes, fmt
        cap_rec, cap_seq = n_bytes // 256 + 16, n_bytes  # the file size bounds the blob; the record guess may grow
        for _ in range(3):
            self.d_seq = torch.empty(cap_seq + 16, dtype=torch.uint8, device=dev)
            self.d_seq_off = torch.zeros(cap_rec + 1, dtype=torch.int64, device=dev)
            self.d_name_start = torch.zeros(cap_rec, dtype=torch.int64, device=dev)
            self.d_name_len = torch.zeros(cap_rec, dtype=torch.int32, device=dev)
            self.d_qual_start = torch.zeros(cap_rec, dtype=torch.int64, device=dev)
            b.d_seq, b.seq_capacity = self.d_seq.data_ptr(), cap_seq
            b.d_seq_off, b.d_name_start = self.d_seq_off.data_ptr(), self.d_name_start.data_ptr()
            b.d_name_len, b.d_qual_start = self.d_name_len.data_ptr(), self.d_qual_start.data_ptr()
            b.max_records = cap_rec
            rc = lib.sa_ingest_device(dev.index or 0, stream.cuda_stream, C.byref(b), ws.data_ptr(), ws.numel())
            if rc != capi.SA_ERR_CAPACITY:
                break
            cap_rec, cap_seq = max(cap_rec, int(b.n_records)), int(b.n_bases)
        capi.check(rc)
        self.n, self.n_bases = int(b.n_records), int(b.n_bases)
        self.min_len, self.max_len = int(b.min_len), int(b.max_len)
        self.d_seq_off = self.d_seq_off[:self.n + 1]
        self.d_name_start, self.d_name_len = self.d_name_start[:self.n], self.d_name_len[:self.n]
        self.d_qual_start = self.d_qual_start[:self.n]

    def lengths(self):
        off = self.d_seq_off.cpu().numpy()
        return (off[1:] - off[:-1]).astype(np.int64)

    def names(self, file_bytes):
        """Names as bytes objects, cut from the host copy of the file."""
        s, l = self.d_name_start.cpu().numpy(), self.d_name_len.cpu().numpy()
        return [bytes(file_bytes[int(a):int(a) + int(b)]) for a, b in zip(s, l)]


def ingest_bytes(data, fmt, device=0):
    """Host bytes -> DeviceRecords (one host-to-device copy of the file as it is)."""
    dev = torch.device("cuda", device)
    host = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
    return DeviceRecords(host.to(dev), fmt, device)


def ingest_file(path, device=0):
    fmt = format_of(path)
    if fmt is None:
        raise ValueError(f"{path}: neither a FASTA nor a FASTQ file name")
    host = torch.from_numpy(np.fromfile(path, dtype=np.uint8))
    return DeviceRecords(host.to(torch.device("cuda", device)), fmt, device)


def split_fragments(lengths, limit=5000):
    """Indices of the short (< limit) and of the long fragments, each in file order - the mapper's
    shortFragments / longFragments (reference src/mapper.cpp:564-577)."""
    lengths = np.asarray(lengths)
    return np.nonzero(lengths < limit)[0], np.nonzero(lengths >= limit)[0]


def stats(lengths, names=None):
    """The figures PrintStats reports (reference src/mapper.cpp:71-105): count, total, longest, shortest,
    average and its "N50": the length of the first fragment, in FILE order, at which the running total
    exceeds half of the total (the reference does not sort by length).  The reference keeps the names of the
    longest / shortest fragment only when they beat the initial values (longest > 0; shortest strictly below
    the first fragment's length, so the first fragment is never named as the shortest)."""
    lengths = [int(x) for x in lengths]
    total = sum(lengths)
    longest, shortest, longest_name, shortest_name = 0, lengths[0], "", ""
    for i, ln in enumerate(lengths):
        if ln > longest:
            longest, longest_name = ln, names[i] if names else ""
        if ln < shortest:
            shortest, shortest_name = ln, names[i] if names else ""
    run, n50 = 0, None
    for ln in lengths:
        run += ln
        if run > 0.5 * total:
            n50 = ln
            break
    return {"count": len(lengths), "total": total, "longest": longest, "longest_name": longest_name,
            "shortest": shortest, "shortest_name": shortest_name, "average": total / len(lengths), "n50": n50}


def pack2bit(d_seq, n_bases, device=0):
    """-> (d_packed int64 [ceil(n/32)], d_other int32 [ceil(n/32)])"""
    dev = torch.device("cuda", device)
    words = (n_bases + 31) // 32
    packed = torch.zeros(max(words, 1), dtype=torch.int64, device=dev)
    other = torch.zeros(max(words, 1), dtype=torch.int32, device=dev)
    capi.check(capi.lib().sa_pack2bit_device(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream,
                                             d_seq.data_ptr(), n_bases, packed.data_ptr(), other.data_ptr()))
    return packed[:words], other[:words]


def unpack2bit(d_packed, d_other, n_bases, device=0):
    dev = torch.device("cuda", device)
    seq = torch.zeros(n_bases + 16, dtype=torch.uint8, device=dev)
    capi.check(capi.lib().sa_unpack2bit_device(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream,
                                               d_packed.data_ptr(), d_other.data_ptr(), n_bases, seq.data_ptr()))
    return seq[:n_bases]


def gather_spans(d_src, starts, lengths, device=0):
    """Cuts spans out of a device blob: -> (uint8 blob of the spans back to back with 8 bytes of slack,
    int64 offsets [n + 1]), both on the device.  starts / lengths are host arrays."""
    dev = torch.device("cuda", device)
    lengths = np.asarray(lengths, dtype=np.int64)
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=off[1:])
    d_off = torch.from_numpy(off).to(dev)
    d_start = torch.from_numpy(np.asarray(starts, dtype=np.int64)).to(dev)
    d_dst = torch.zeros(int(off[-1]) + 16, dtype=torch.uint8, device=dev)
    capi.check(capi.lib().sa_gather_spans_device(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream,
                                                 d_src.data_ptr(), d_start.data_ptr(), d_off.data_ptr(),
                                                 len(lengths), d_dst.data_ptr()))
    return d_dst, d_off
