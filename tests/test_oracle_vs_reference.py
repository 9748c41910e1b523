"""CPU suite: differential test of the restatement against the compiled reference (oracle/_ref),
on seeded random inputs inside the reference's defined-behaviour envelope.  Skipped where the
reference build is absent."""
import random

import numpy as np

from sequencealigner_amd import synth

SCHEMES = [(1, -1, -1), (2, -2, -1), (0, -1, -1), (2, -1, -1), (1, -3, -2), (1, 1, -1), (1, -1, 0),
           (1, -1, 1)]


def test_align_differential(oracle, reference_lib):
    rng = random.Random(77)
    n = 0
    for _ in range(1500):
        alpha = rng.choice(["ACGT", "AC", "ACGTN", "A"])
        t = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 70)))
        if rng.random() < 0.5:
            cut = t[rng.randint(0, len(t) // 2):]
            q = "".join(c if rng.random() > 0.15 else rng.choice(alpha) for c in cut) or "A"
        else:
            q = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 70)))
        for m, x, g in SCHEMES:
            for typ in (0, 1, 2):
                r = oracle.align(q, t, typ, m, x, g)
                if typ == 1 and r[0] <= 0:
                    continue   # reference reads uninitialised maxRow/maxCol (alignment.cpp:171,212)
                if typ == 2 and r[0] <= 0 and g <= 0:
                    continue   # reference reads uninitialised maxRow (alignment.cpp:56,98)
                assert r == reference_lib.align(q, t, typ, m, x, g), (q, t, typ, m, x, g)
                n += 1
    assert n > 20000


def test_minimize_binary_differential(oracle, reference_lib):
    rng = random.Random(78)
    for _ in range(1200):
        k = rng.randint(1, 16)
        w = rng.randint(1, k + 1)
        L = rng.randint(k + w - 1, k + w - 1 + rng.choice([0, 1, 5, 50, 400]))
        alpha = rng.choice(["ACGT", "acgt", "AC", "A", "ACGTacgt"])
        if rng.random() < 0.3:
            unit = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 6)))
            s = (unit * (L // len(unit) + 1))[:L]
        else:
            s = "".join(rng.choice(alpha) for _ in range(L))
        assert oracle.minimize_binary(s, k, w) == reference_lib.minimize_binary(s, k, w), (s, k, w)


def test_minimize_binary_other_bytes_differential(oracle, reference_lib):
    """Bytes that are not bases get the code -1 in the reference (src/minimizer_binary.cpp:28-29, 46-47): the
    restatement must follow it there too (this pins the checker the GPU tests use for such input)."""
    rng = random.Random(31)
    for _ in range(1500):
        k = rng.randint(1, 16)
        w = rng.randint(1, k + 1)
        L = rng.randint(k + w - 1, 300)
        alpha = rng.choice(["ACGTN", "ACGTNacgtn", "ACGTRYKM", "ACGT" * 10 + "N"])
        s = "".join(rng.choice(alpha) for _ in range(L))
        if rng.random() < 0.3:
            a = rng.randint(0, L - 1)
            b = min(L, a + rng.randint(1, 40))
            s = s[:a] + "N" * (b - a) + s[b:]
        assert oracle.minimize_binary(s, k, w) == reference_lib.minimize_binary(s, k, w), (s, k, w)


def test_minimize_binary_long_differential(oracle, reference_lib):
    g = bytes(synth.genome(300000, seed=31)) + b"T" * 5000 + bytes(synth.genome(1000, seed=32))
    a = oracle.minimize_binary_array(g, 15, 5)
    b = reference_lib.minimize_binary_array(g, 15, 5)
    assert np.array_equal(a, b)


def test_batch_entry_points_agree(oracle, reference_lib):
    q, t = synth.pairs_short_read_window(64, seed=41)
    qb, qo = synth.to_blob(q)
    tb, to = synth.to_blob(t)
    s1, b1 = oracle.align_batch(qb, qo, tb, to, 1, 1, -1, -1, True, 4)
    s2, b2 = reference_lib.align_batch(qb, qo, tb, to, 1, 1, -1, -1, True, 4)
    assert np.array_equal(s1, s2) and np.array_equal(b1, b2)
    for i in range(8):
        one = oracle.align(bytes(q[i]), bytes(t[i]), 1, 1, -1, -1)
        assert one[0] == s1[i] and one[2] == b1[i]


def test_linear_memory_scorer_differential(oracle, reference_lib):
    """The two-row scorer (used to check pairs whose full matrix does not fit in host memory) returns the
    score of the compiled reference's Align(): small pairs over every scheme and mode, then pairs of up to
    5 kbp of the benchmark generators."""
    rng = random.Random(79)
    n = 0
    for _ in range(600):
        alpha = rng.choice(["ACGT", "AC", "ACGTN", "A"])
        t = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 90)))
        if rng.random() < 0.5:
            cut = t[rng.randint(0, len(t) // 2):]
            q = "".join(c if rng.random() > 0.15 else rng.choice(alpha) for c in cut) or "A"
        else:
            q = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 90)))
        for m, x, g in SCHEMES:
            for typ in (0, 1, 2):
                got = oracle.score_linear(q, t, typ, m, x, g)
                assert got == oracle.align(q, t, typ, m, x, g)[0]
                if typ != 0 and got <= 0:
                    continue   # outside the reference's defined behaviour (uninitialised end cell: it may crash)
                want = reference_lib.align(q, t, typ, m, x, g, want_cigar=False, want_target_begin=False)[0]
                assert got == want, (q, t, typ, m, x, g)
                n += 1
    assert n > 9000
    qq, tt = synth.pair_global_long(5000, seed=12)
    qs, ts = synth.pairs_semiglobal_long(2, seed=13, read_len=3000, window_len=3300, lead=150)
    for q, t in ((bytes(qq), bytes(tt[:4700])), (bytes(qs[0]), bytes(ts[0])), (bytes(qs[1]), bytes(ts[1]))):
        for typ, (m, x, g) in ((0, (1, -1, -1)), (2, (1, -1, -1)), (1, (2, -3, -2)), (0, (1, -1, 0))):
            want = reference_lib.align(q, t, typ, m, x, g, want_cigar=False, want_target_begin=False)[0]
            assert oracle.score_linear(q, t, typ, m, x, g) == want


def test_minimize_binary_k_above_16_differential(oracle, reference_lib):
    """k > 16: the reference's 32-bit k-mer values wrap (src/minimizer_binary.cpp:53-78: the shifts drop the
    high bits, getMask saturates at 0xFFFFFFFF), so a value is the last 16 bases of the k-mer, while positions,
    the expiry rule and the window count still use k.  The restatement must follow the compiled reference."""
    rng = random.Random(80)
    for _ in range(1200):
        k = rng.randint(17, 31)
        w = rng.randint(1, min(k + 1, 24))
        L = rng.randint(k + w - 1, k + w - 1 + rng.choice([0, 1, 5, 50, 400]))
        alpha = rng.choice(["ACGT", "acgt", "AC", "A", "ACGTacgt", "ACGTN", "ACGT" * 10 + "N"])
        if rng.random() < 0.3:
            unit = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 6)))
            s = (unit * (L // len(unit) + 1))[:L]
        else:
            s = "".join(rng.choice(alpha) for _ in range(L))
        assert oracle.minimize_binary(s, k, w) == reference_lib.minimize_binary(s, k, w), (s, k, w)
