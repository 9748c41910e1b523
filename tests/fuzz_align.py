"""Randomised parity run of batched Align() against the CPU restatement: boundary-heavy lengths (around the
strip, tile and lane-block sizes of the kernels), random scoring schemes inside the reference's defined
envelope, small alphabets, every kernel family.  Prints one line per batch and stops at the first mismatch.

    python tests/fuzz_align.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cpu_checker  # noqa: E402
from sequencealigner_amd import align, capi  # noqa: E402

EDGES = [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 95, 96, 97, 127, 128, 129, 255, 256,
         257, 300, 511, 512, 513, 1023, 1024, 1025, 1500, 2047, 2048, 2049]


def length(rng, cap):
    x = rng.choice(EDGES) if rng.random() < 0.6 else rng.randint(1, cap)
    return min(x, cap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    oracle = cpu_checker.restatement()
    t_end = time.time() + args.seconds
    batches = pairs_checked = 0
    while time.time() < t_end:
        fam = rng.choice([None, "lane", "wave", "wave32", "coop", "coop"])
        capi.set_option("force_i32", 1 if fam == "wave32" else 0)
        capi.set_option("force_family", {None: 0, "lane": 1}.get(fam, 2))
        capi.set_option("force_coop", 1 if fam == "coop" else 0 if fam in ("wave", "wave32") else -1)
        # the cooperative family: half of the batches with 64-row traceback segments (pairs of a few hundred rows
        # then take the many-wavefront traceback), with 0..3 speculative walks and sometimes no room for the runs
        short_segments = fam == "coop" and rng.random() < 0.5
        capi.set_option("seg_rows", 64 if short_segments else 0)
        capi.set_option("seg_spec", rng.choice([0, 1, 2, 2, 3]) if short_segments else 2)
        capi.set_option("seg_cap", 8 if short_segments and rng.random() < 0.2 else 0)
        # and with few wavefronts per pass, so that a batch takes several passes
        capi.set_option("coop_waves", rng.choice([64, 256, 8192]) if fam == "coop" else 8192)
        # in the packed form where it applies, its strips in one piece or in segments of 64 .. 256 target rows
        capi.set_option("coop_pk", rng.choice([-1, 0, 1, 1]) if fam == "coop" else -1)
        capi.set_option("coop_seg_rows", rng.choice([-1, 0, 64, 64, 128, 256]) if fam == "coop" else -1)
        typ = rng.choice([0, 1, 2])
        match, mismatch, gap = rng.choice([0, 1, 1, 2, 3]), rng.choice([-3, -2, -1, -1, 0, 1]), rng.choice([-3, -2, -1, -1, 0, 1])
        big = rng.random() < 0.15
        cap = 2100 if big else 400
        n = rng.randint(1, 6 if big else 200)
        alpha = rng.choice(["ACGT", "ACGT", "AC", "ACGTN", "A"])
        pairs = []
        for _ in range(n):
            T = length(rng, cap)
            t = "".join(rng.choice(alpha) for _ in range(T))
            if rng.random() < 0.7:
                a = rng.randint(0, T // 2)
                cut = t[a:a + length(rng, max(T - a, 1))]
                q = "".join(c if rng.random() > 0.1 else rng.choice(alpha) for c in cut) or alpha[0]
            else:
                q = "".join(rng.choice(alpha) for _ in range(length(rng, cap)))
            pairs.append((q, t))
        s, c, tb = align.align_batch([p[0] for p in pairs], [p[1] for p in pairs], typ, match, mismatch, gap)
        for i, (q, t) in enumerate(pairs):
            want = oracle.align(q, t, typ, match, mismatch, gap)
            if typ == 1 and want[0] <= 0:
                continue  # the reference reads uninitialised indices when no cell is positive
            if typ == 2 and want[0] <= 0 and gap <= 0:
                continue
            got = (int(s[i]), c[i], int(tb[i]))
            if got != want:
                print("MISMATCH", fam, typ, match, mismatch, gap, repr(q), repr(t), got, want, flush=True)
                sys.exit(1)
            pairs_checked += 1
        batches += 1
        if batches % 20 == 0:
            print(f"{batches} batches, {pairs_checked} pairs equal", flush=True)
    print(f"done: {batches} batches, {pairs_checked} pairs equal", flush=True)


if __name__ == "__main__":
    main()
