"""Randomised parity runs of the minimizer and ingest paths against the CPU restatement.

Minimizers: random (k, w), lengths around the chunk and warm-up sizes of the kernels, tandem repeats and
homopolymers (which defeat the warm-up), bytes other than bases, batches of mixed lengths.
Ingest: FASTA / FASTQ files from the generators of the tests, then damaged at random (bytes overwritten,
inserted or removed, truncated): both sides must accept the same files and agree on every output.

    python tests/fuzz_other.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cpu_checker  # noqa: E402
import seqfiles  # noqa: E402
from sequencealigner_amd import capi, ingest, minimize  # noqa: E402

EDGES = [0, 1, 2, 7, 8, 9, 63, 64, 65, 159, 160, 161, 255, 256, 257, 415, 416, 417, 1023, 1024, 1025, 2047, 2048, 2049,
         4095, 4096, 4097, 9000, 20000]


def minimizer_round(rng, oracle):
    k = rng.randint(1, 16)
    w = rng.choice([1, rng.randint(1, k + 1), min(5, k + 1)])
    seqs = []
    for _ in range(rng.randint(1, 40)):
        extra = rng.choice(EDGES) if rng.random() < 0.7 else rng.randint(0, 3000)
        L = k + w - 1 + extra
        alpha = rng.choice(["ACGT", "ACGT", "acgt", "AC", "A", "ACGTN"])
        mode = rng.random()
        if mode < 0.25:
            unit = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 12)))
            s = (unit * (L // len(unit) + 1))[:L]
        elif mode < 0.35:
            s = list("".join(rng.choice("ACGT") for _ in range(L)))
            a = rng.randint(0, L - 1)
            for x in range(a, min(L, a + rng.choice([1, 5, 50, 500]))):
                s[x] = "N"
            s = "".join(s)
        else:
            s = "".join(rng.choice(alpha) for _ in range(L))
        seqs.append(s)
    res = minimize.minimize_batch(seqs, k, w)
    for s, r in zip(seqs, res):
        want = oracle.minimize_binary_array(s, k, w)
        if not np.array_equal(r, want):
            print("MINIMIZER MISMATCH", k, w, len(s), repr(s[:200]), flush=True)
            sys.exit(1)
    return len(seqs)


def damage(rng, data):
    b = bytearray(data)
    for _ in range(rng.randint(0, 3)):
        if not b:
            break
        x = rng.randint(0, len(b) - 1)
        what = rng.random()
        if what < 0.3:
            b[x] = rng.choice(b">@+\n\r AN")
        elif what < 0.5:
            b.insert(x, rng.choice(b">@+\n\r A"))
        elif what < 0.7:
            del b[x]
        elif what < 0.85:
            del b[x:]
        else:
            del b[x:x + rng.randint(1, 200)]
    return bytes(b)


def ingest_round(rng, nprng, oracle):
    fmt = rng.choice([0, 1])
    lengths = [rng.choice([1, 2, 59, 60, 61, 1000, 16000, 17000]) if rng.random() < 0.5 else rng.randint(1, 3000)
               for _ in range(rng.randint(1, 30))]
    eol = rng.choice([b"\n", b"\n", b"\r\n"])
    if fmt == 0:
        data, _, _ = seqfiles.fasta(nprng, lengths, width=rng.choice([0, 1, 7, 60, 80]), eol=eol,
                                    last_eol=rng.random() < 0.7, header_len=rng.choice([0, 0, 300, 20000]))
    else:
        data, _, _ = seqfiles.fastq(nprng, lengths, eol=eol, last_eol=rng.random() < 0.7)
    if rng.random() < 0.6:
        data = damage(rng, data)
    want = oracle.parse(data, fmt)
    try:
        rec = ingest.ingest_bytes(data, fmt)
    except capi.SaError as e:
        if e.code != capi.SA_ERR_FORMAT or want is not None:
            print("INGEST MISMATCH (rejected)", fmt, e, repr(data[:300]), flush=True)
            sys.exit(1)
        return 0
    if want is None:
        print("INGEST MISMATCH (accepted)", fmt, repr(data[:300]), flush=True)
        sys.exit(1)
    seq, off, ns, nl, qs = want
    ok = (rec.n == len(off) - 1 and np.array_equal(rec.d_seq_off.cpu().numpy().astype(np.uint64), off)
          and np.array_equal(rec.d_seq[:rec.n_bases].cpu().numpy(), seq)
          and np.array_equal(rec.d_name_start.cpu().numpy().astype(np.uint64), ns)
          and np.array_equal(rec.d_name_len.cpu().numpy().astype(np.uint32), nl)
          and (fmt == 0 or np.array_equal(rec.d_qual_start.cpu().numpy().astype(np.uint64), qs)))
    if not ok:
        print("INGEST MISMATCH (outputs)", fmt, repr(data[:300]), flush=True)
        sys.exit(1)
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    nprng = np.random.default_rng(args.seed)
    oracle = cpu_checker.restatement()
    t_end = time.time() + args.seconds
    rounds = seqs = files = rejected = 0
    while time.time() < t_end:
        seqs += minimizer_round(rng, oracle)
        for _ in range(3):
            got = ingest_round(rng, nprng, oracle)
            files += 1
            rejected += 1 - got
        rounds += 1
        if rounds % 20 == 0:
            print(f"{rounds} rounds: {seqs} sequences, {files} files ({rejected} rejected by both) equal", flush=True)
    print(f"done: {seqs} sequences, {files} files ({rejected} rejected by both) equal", flush=True)


if __name__ == "__main__":
    main()
