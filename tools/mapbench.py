"""The mapper's main path end to end on the GPU, stage by stage: FASTA + FASTQ bytes -> ingest -> minimizers ->
chaining -> cut the chained spans -> Global alignment -> CIGAR statistics (reference src/mapper.cpp:532-679 for
reads below 5 000 bases with -c).  Prints one JSON line with the time of every stage; the PAF text itself
(host formatting) is left out.

    python tools/mapbench.py [--reads 20000] [--length 3000] [--genome 4600000]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sequencealigner_amd import align, chain, ingest, minimize, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--length", type=int, default=3000)
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--error", type=float, default=0.05)
    args = ap.parse_args()
    rng = np.random.default_rng(5)
    g = np.asarray(synth.genome(args.genome, seed=4), dtype=np.uint8)
    starts = rng.integers(0, args.genome - args.length, args.reads)
    reads = g[starts[:, None] + np.arange(args.length)[None, :]].copy()
    flip = rng.random(reads.shape) < args.error
    reads[flip] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(flip.sum()))]
    fasta = b">chr synthetic\n" + b"\n".join(bytes(g[i:i + 80]) for i in range(0, args.genome, 80)) + b"\n"
    qual = b"I" * args.length
    fastq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(reads[i]), qual) for i in range(args.reads))
    d_fa = torch.frombuffer(bytearray(fasta), dtype=torch.uint8).cuda()
    d_fq = torch.frombuffer(bytearray(fastq), dtype=torch.uint8).cuda()
    k, w, f = 15, 5, 0.001
    best = None
    for _ in range(3):
        t = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = ingest.DeviceRecords(d_fa, ingest.FASTA)
        rd = ingest.DeviceRecords(d_fq, ingest.FASTQ)
        torch.cuda.synchronize()
        t["ingest"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        mr = minimize.DeviceMinimizeBatch.from_device(ref.d_seq, ref.d_seq_off, 1, ref.n_bases, ref.n_bases, ref.n_bases, k, w)
        mr.run()
        mq = minimize.DeviceMinimizeBatch.from_device(rd.d_seq, rd.d_seq_off, rd.n, rd.n_bases, rd.min_len, rd.max_len, k, w)
        mq.run()
        torch.cuda.synchronize()
        n_ref = int(mr.d_out_off[-1].item())
        n_rd = int(mq.d_out_off[-1].item())
        t["minimizers"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        spans = chain.chain_batch(mr.d_out[:3 * n_ref].reshape(n_ref, 3), mq.d_out[:3 * n_rd].reshape(n_rd, 3),
                                  mq.d_out_off, f)
        t["chain"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        lis, tb, te, qb, qe = (spans[:, c].astype(np.int64) for c in range(5))
        idx = np.nonzero((lis > 0) & ((te - tb) < 100000) & (qe > qb) & (te > tb))[0]
        off = rd.d_seq_off.cpu().numpy()
        classes, has_rest = align.size_classes(qe[idx] - qb[idx], te[idx] - tb[idx])
        cells = matches = block = 0
        kernels = {}
        side = torch.cuda.Stream() if has_rest else None
        batches = []
        for n_cls, cls in enumerate(classes):  # one batch per size class, the rare sizes last
            sel = idx[cls]
            d_q, d_qo = ingest.gather_spans(rd.d_seq, off[sel] + qb[sel], qe[sel] - qb[sel])
            d_t, d_to = ingest.gather_spans(ref.d_seq, tb[sel], te[sel] - tb[sel])
            batches.append(align.DeviceAlignBatch.from_device(d_q, d_qo, d_t, d_to, qe[sel] - qb[sel],
                                                              te[sel] - tb[sel], align.AlignmentType.Global, 1, -1, -1))
        torch.cuda.synchronize()
        t["cut spans + buffers"] = time.perf_counter() - t0
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        # few pairs of rare sizes: on a stream of their own beside the large classes, their call in a thread of its
        # own (calls that take the cooperative kernels wait for them; see sequencealigner_amd/mapper.py)
        worker = None
        if side is not None:
            worker = threading.Thread(target=batches[-1].run, args=(side,))
            worker.start()
        for batch in (batches[:-1] if side is not None else batches):
            batch.run()
        if worker is not None:
            worker.join()
        torch.cuda.synchronize()
        t["align"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for batch in batches:
            bl, ops = batch.cigar_stats()
            cells += int(batch.cells)
            matches += int(ops[:, 0].sum())
            block += int(bl.sum())
            kernels[batch.kernel_name()] = kernels.get(batch.kernel_name(), 0) + batch.n
        t["cigar stats"] = time.perf_counter() - t0
        del batches
        t["total"] = sum(t.values())
        if best is None or t["total"] < best["total"]:
            best = dict(t)
            info = {"aligned": int(idx.size), "size_classes": len(classes), "pairs_by_kernel": kernels, "cells": cells,
                    "mean_identity": matches / max(block, 1)}
    out = {"case": f"{args.reads} reads x {args.length} bp, {args.error:.0%} substitutions, genome {args.genome} bp",
           "ms": {k2: round(v * 1e3, 3) for k2, v in best.items()}, **info,
           "reads_per_s": round(args.reads / best["total"]),
           "align_GCUPS": round(info["cells"] / best["align"] / 1e9, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
