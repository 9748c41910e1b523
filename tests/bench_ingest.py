"""Throughput of the device FASTA / FASTQ ingest on the shapes of the minimizer workload: a 4.6 Mbp genome
as one FASTA record (80 columns) and 10 kbp reads as FASTQ, 1 Gbp in all.  Prints one JSON line per case:
whole-call time (two kernels, a scan and two stream synchronisations), file GB/s, and the CPU restatement's
single-thread rate on a sample of the same bytes.

    python tests/bench_ingest.py [--reads 100000] [--cpu-mb 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sequencealigner_amd import ingest  # noqa: E402
import seqfiles  # noqa: E402


def timed(d_file, fmt, reps):
    ingest.DeviceRecords(d_file, fmt)  # warm-up (allocations, code load)
    torch.cuda.synchronize()
    best, rec = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        rec = ingest.DeviceRecords(d_file, fmt)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--cpu-mb", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    import cpu_checker
    cpu = cpu_checker.restatement()

    cases = []
    genome, _, _ = seqfiles.fasta(rng, [4_600_000], width=80, name_extra=b" synthetic genome")
    cases.append(("fasta_genome_4.6Mbp", genome, 0, 1))
    block_reads = 2000  # one block of distinct reads, repeated on the device to the full size
    block, _, _ = seqfiles.fastq(rng, [10_000] * block_reads)
    cases.append(("fastq_reads_10kbp", block, 1, max(args.reads // block_reads, 1)))

    for name, data, fmt, repeat in cases:
        host = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        d_file = host.cuda().repeat(repeat)
        n_bytes = int(d_file.numel())
        t, rec = timed(d_file, fmt, args.reps)
        sample = data[:args.cpu_mb << 20]
        if fmt == 1:  # cut at a record boundary
            sample = sample[:sample.rfind(b"\n@read") + 1]
        t0 = time.perf_counter()
        got = cpu.parse(sample, fmt)
        t_cpu = time.perf_counter() - t0
        assert got is not None
        # algorithmic bytes: the file read once, the bases written once, 28 bytes of tables per record
        algo = n_bytes + rec.n_bases + 28 * rec.n
        print(json.dumps({
            "case": name, "file_bytes": n_bytes, "records": rec.n, "bases": rec.n_bases,
            "ms_per_call": round(t * 1e3, 3), "file_GBps": round(n_bytes / t / 1e9, 1),
            "algorithmic_GBps": round(algo / t / 1e9, 1), "hbm_frac_of_8TBps": round(algo / t / 8e12, 3),
            "cpu_restatement_GBps_1_thread": round(len(sample) / t_cpu / 1e9, 3),
            "cpu_sample_bytes": len(sample)}), flush=True)


if __name__ == "__main__":
    main()
