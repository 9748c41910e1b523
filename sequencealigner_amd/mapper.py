"""The mapper's per-read pipeline (mapping(), reference src/mapper.cpp:401-444) as four batched GPU calls:
minimizers of all reads, seed chaining against the reference's minimizers, Global alignment of the chained
spans, CIGAR statistics - then one PAF record per read (generatePAFString, :354-369).

Branches of mapping() (:421-442): a read with a chain, a chained reference span below 100 000 and a length
below 5 000 is aligned when CIGARs are wanted; otherwise a record without CIGAR reports chain length * k and
the larger of the two spans; a read without a chain gets the "-1" record when CIGARs are wanted."""
import numpy as np
import torch

from . import align, chain, minimize, paf


def _device_lists(seqs, k, w, device):
    b = minimize.DeviceMinimizeBatch(seqs, k, w, device=device)
    b.run()
    torch.cuda.synchronize(device)
    off = b.d_out_off.clone()
    n = int(off[-1].item())
    return b.d_out[:3 * n].clone().reshape(n, 3), off


def map_reads(ref_name, ref_seq, read_names, reads, k=15, w=5, f=0.001, match=1, mismatch=-1, gap=-1,
              cigar_needed=True, device=0):
    """Returns the PAF records (strings ending in a newline) in read order."""
    ref_seq = bytes(ref_seq)
    reads = [bytes(r) for r in reads]
    d_ref, _ = _device_lists([ref_seq], k, w, device)
    d_reads, d_off = _device_lists(reads, k, w, device)
    spans = chain.chain_batch(d_ref, d_reads, d_off, f, device=device)   # length, t_begin, t_end, q_begin, q_end
    n = len(reads)
    lens = np.array([len(r) for r in reads])
    lis, tb, te, qb, qe = (spans[:, c] for c in range(5))
    to_align = (lis > 0) & ((te - tb) < 100000) & cigar_needed & (lens < 5000)
    idx = np.nonzero(to_align)[0]
    cigars, block = {}, {}
    if idx.size:
        batch = align.DeviceAlignBatch([reads[i][qb[i]:qe[i]] for i in idx], [ref_seq[tb[i]:te[i]] for i in idx],
                                       align.AlignmentType.Global, match, mismatch, gap, device=device)
        batch.run()
        torch.cuda.synchronize(device)
        bl, _ = batch.cigar_stats()
        for j, (i, c) in enumerate(zip(idx, batch.cigars())):
            cigars[i], block[i] = c, int(bl[j])
    out = []
    for i in range(n):
        if to_align[i]:
            # the mapper's match length looks for a lower-case 'm' and is always 0 (src/mapper.cpp:392)
            out.append(paf.paf_line(read_names[i], lens[i], qb[i], qe[i], ref_name, len(ref_seq), tb[i], te[i], 0,
                                    block[i], cigars[i]))
        elif (not cigar_needed) or (te[i] - tb[i]) >= 100000 or lens[i] >= 5000:
            out.append(paf.paf_line(read_names[i], lens[i], qb[i], qe[i], ref_name, len(ref_seq), tb[i], te[i],
                                    int(lis[i]) * k, max(int(qe[i] - qb[i]), int(te[i] - tb[i]))))
        else:
            out.append(paf.paf_line(read_names[i], lens[i], -1, -1, ref_name, len(ref_seq), -1, -1, 0, 0))
    return out
