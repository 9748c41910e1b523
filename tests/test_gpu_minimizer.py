"""GPU parity tests for batched MinimizeBinary()/Minimize(): the HIP path (through the C-ABI) against
the golden fixtures produced by the compiled reference and against the CPU restatement.  Bit-exact bar:
identical (value, position, strand) lists in identical order."""
import hashlib
import random

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu


def _tl(arr):
    return [tuple(int(x) for x in r) for r in arr]


def test_reference_unit_test_answers():
    from sequencealigner_amd.minimize import Minimize, MinimizeBinary
    # test/testAlignmentMinimizer.cpp:37-57 and :61-82
    assert MinimizeBinary("TGACGTACATGGACA", 15, 3, 4) == [(2, 10, False), (6, 4, False), (6, 7, True), (11, 1, False)]
    assert Minimize("TGACGTACATGGACA", 15, 3, 4) == [(2, 10, False), (12, 4, False), (12, 7, True), (23, 1, False)]


def test_golden_cases_one_batch_per_kw():
    from sequencealigner_amd import capi, minimize
    cases = golden_io.load("minimizer_cases.json")
    groups = {}
    for c in cases:
        groups.setdefault((c["variant"], c["k"], c["w"]), []).append(c)
    n = 0
    for (variant, k, w), cs in groups.items():
        v = capi.SA_MIN_BINARY if variant == "binary" else capi.SA_MIN_DECIMAL
        throwers = [c for c in cs if c["out"] == "throws"]
        ok = [c for c in cs if c["out"] != "throws"]
        if ok:
            res = minimize.minimize_batch([c["seq"] for c in ok], k, w, v)
            for c, r in zip(ok, res):
                assert _tl(r) == [tuple(x) for x in c["out"]], (c["seq"], k, w, variant)
                n += 1
        for c in throwers:
            with pytest.raises(OverflowError):
                minimize.Minimize(c["seq"], len(c["seq"]), k, w)
            n += 1
    assert n == len(cases)


def test_golden_long_sequences():
    from sequencealigner_amd import minimize
    for c in golden_io.load("minimizer_long.json"):
        arr = minimize.minimize_binary(golden_io.long_sequence(c["name"]), c["k"], c["w"])
        assert arr.shape[0] == c["count"], c["name"]
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == c["sha256"], c["name"]


def test_random_vs_oracle(oracle):
    from sequencealigner_amd import minimize
    rng = random.Random(99)
    for _ in range(40):
        k = rng.randint(1, 16)
        w = rng.randint(1, k + 1)
        seqs = []
        for _ in range(rng.randint(1, 60)):
            L = rng.randint(k + w - 1, k + w - 1 + rng.choice([0, 3, 100, 3000, 7000]))
            alpha = rng.choice(["ACGT", "ACGT", "acgt", "AC", "A"])
            if rng.random() < 0.3:
                unit = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 9)))
                s = (unit * (L // len(unit) + 1))[:L]
            else:
                s = "".join(rng.choice(alpha) for _ in range(L))
            seqs.append(s)
        res = minimize.minimize_batch(seqs, k, w)
        for s, r in zip(seqs, res):
            assert np.array_equal(r, oracle.minimize_binary_array(s, k, w)), (k, w, len(s))


def test_genome_and_reads_shapes_vs_oracle(oracle):
    from sequencealigner_amd import minimize, synth
    g = bytes(synth.genome(1_000_000, seed=71)) + b"G" * 40000 + bytes(synth.genome(200_000, seed=72)) \
        + b"ACT" * 20000 + bytes(synth.genome(100_000, seed=73))
    got = minimize.minimize_binary(g, 15, 5)
    want = oracle.minimize_binary_array(g, 15, 5)
    assert np.array_equal(got, want)
    reads = synth.reads(300, 10_000, seed=74)
    res = minimize.minimize_batch(reads, 15, 5)
    for i in range(300):
        assert np.array_equal(res[i], oracle.minimize_binary_array(bytes(reads[i]), 15, 5)), i


def test_benchmark_inputs_every_triple_vs_oracle(oracle):
    """The benchmark's own minimizer inputs at full size (bench.py::minimizer_leg): the 4.6 Mbp genome compared
    triple for triple, and all 100 000 reads of 10 kbp compared through the count and the hash of every read's
    list (the CPU checker hashes the list it computed, cpu_checker.triple_list_hashes the device result), with
    every 997th read's list compared in full."""
    import cpu_checker
    from sequencealigner_amd import minimize, synth
    g = bytes(synth.genome(4_600_000, seed=4))
    assert np.array_equal(minimize.minimize_binary(g, 15, 5), oracle.minimize_binary_array(g, 15, 5))
    reads = synth.reads(100_000, 10_000, seed=6)
    n, length = reads.shape
    b = minimize.DeviceMinimizeBatch(reads, 15, 5)
    b.run()
    assert b.status() == 0
    got_hash = cpu_checker.triple_list_hashes(b.d_out, b.d_out_off)
    got_count = b.counts()
    blob = np.concatenate([reads, np.zeros((n, 1), np.uint8)], axis=1)   # NUL-terminated, as the reference reads them
    off = np.arange(n, dtype=np.uint64) * np.uint64(length + 1)
    want_count, want_hash = oracle.minimize_batch(blob.reshape(-1), off, np.full(n, length, np.uint32), 15, 5, 16)
    assert np.array_equal(got_count.astype(np.uint64), want_count)
    assert np.array_equal(got_hash, want_hash)
    res = b.results()
    for i in range(0, n, 997):
        assert np.array_equal(res[i], oracle.minimize_binary_array(bytes(reads[i]), 15, 5)), i


def test_sortedness_and_uniqueness_property():
    """Size-independent property on a larger batch: every output list is strictly increasing in
    (value, position, strand) and every position lies inside its sequence."""
    from sequencealigner_amd import minimize, synth
    reads = synth.reads(2000, 10_000, seed=75)
    res = minimize.minimize_batch(reads, 15, 5)
    for r in res[::97]:
        key = (r[:, 0].astype(np.uint64) << np.uint64(33)) | (r[:, 1].astype(np.uint64) << np.uint64(1)) | r[:, 2].astype(np.uint64)
        assert (np.diff(key.astype(np.int64)) > 0).all()
        assert r[:, 1].max() <= 10_000 - 15


@pytest.mark.parametrize("chunk", ["8", "24", "64", "1024"])
def test_mixed_lengths_and_chunk_sizes_vs_oracle(oracle, chunk):
    """Sequences with more keys than the in-LDS sort takes (left to the segmented sort) next to short
    ones, constant and periodic stretches that keep the candidate sets from collapsing, and chunk
    lengths below / equal to / above the warm-up length."""
    from sequencealigner_amd import capi, minimize, synth
    capi.set_option("min_chunk", int(chunk))
    try:
        _mixed_lengths(oracle, chunk, minimize, synth)
    finally:
        capi.set_option("min_chunk", 0)


def _mixed_lengths(oracle, chunk, minimize, synth):
    rng = random.Random(5)
    seqs = [bytes(synth.genome(40_000, seed=81)),
            bytes(synth.genome(3_000, seed=82)) + b"A" * 5000 + bytes(synth.genome(9_000, seed=83)) + b"CG" * 3000,
            b"ACGTTGCA" * 2500, b"ACGTTGCAAC" * 500, b"T" * 30000, bytes(synth.genome(32, seed=84))]
    seqs += [bytes(synth.genome(rng.randint(32, 12_000), seed=90 + i)) for i in range(40)]
    for k, w in ((15, 5), (4, 5), (16, 17), (2, 1)):
        res = minimize.minimize_batch(seqs, k, w)
        for s, r in zip(seqs, res):
            assert np.array_equal(r, oracle.minimize_binary_array(s, k, w)), (k, w, len(s), chunk)


def _with_other_bytes(rng, L, alpha):
    s = [rng.choice(alpha) for _ in range(L)]
    for _ in range(rng.randint(1, 4)):           # runs of N, some long, some at the very start
        a = rng.choice([0, rng.randint(0, L - 1)])
        for x in range(a, min(L, a + rng.choice([1, 1, 2, 7, 40, 400]))):
            s[x] = "N"
    return "".join(s)


def test_bytes_other_than_bases_vs_oracle(oracle):
    """The reference gives every byte that is not one of ACGT / acgt the code -1, which sets all bits of the
    k-mer value (src/minimizer_binary.cpp:28-29, 46-47, 62-68), and builds k-mer 0 without the mask (:53-60).
    The restatement agrees with the compiled reference on such input (tests/test_oracle_vs_reference.py);
    here the GPU path must agree with the restatement: clean and unclean sequences in one batch, w = 1,
    k = 16, long runs, chunked long sequences."""
    from sequencealigner_amd import minimize
    rng = random.Random(1234)
    for _ in range(30):
        k = rng.randint(1, 16)
        w = rng.choice([1, rng.randint(1, k + 1)])
        seqs = []
        for _ in range(rng.randint(1, 40)):
            L = rng.randint(k + w - 1, k + w - 1 + rng.choice([0, 5, 200, 3000, 9000]))
            alpha = rng.choice(["ACGT", "ACGTN", "acgtn", "ACGTRYKMSW*-", "AC"])
            s = "".join(rng.choice(alpha) for _ in range(L))
            if rng.random() < 0.5:
                s = _with_other_bytes(rng, L, "ACGT")
            seqs.append(s)
        res = minimize.minimize_batch(seqs, k, w)
        for s, r in zip(seqs, res):
            assert np.array_equal(r, oracle.minimize_binary_array(s, k, w)), (k, w, len(s), s[:80])


def test_long_sequence_with_gaps_of_n_vs_oracle(oracle):
    from sequencealigner_amd import minimize, synth
    g = bytearray(synth.genome(700_000, seed=91))
    for a, n in ((0, 30), (5_000, 1), (123_456, 20_000), (400_000, 3), (400_020, 1), (699_990, 10)):
        g[a:a + n] = b"N" * n
    g = bytes(g)
    for k, w in ((15, 5), (16, 17), (8, 1)):
        got = minimize.minimize_binary(g, k, w)
        assert np.array_equal(got, oracle.minimize_binary_array(g, k, w)), (k, w)


def test_k_above_16_vs_oracle(oracle):
    """k = 17 .. 31 (mapper settings such as -k 19 or -k 21): the reference's 32-bit k-mer values wrap, a value
    is the last 16 bases of the k-mer (src/minimizer_binary.cpp:53-78) while positions, expiry and the window
    count use k.  Random reads, constant and periodic stretches, bytes other than bases, one long sequence."""
    from sequencealigner_amd import minimize, synth
    rng = random.Random(23)
    seqs = [bytes(synth.genome(rng.randint(64, 9_000), seed=400 + i)) for i in range(60)]
    seqs += [b"A" * 3000, b"ACGTTGCA" * 700, bytes(synth.genome(2000, seed=7)) + b"N" * 40 + bytes(synth.genome(3000, seed=8)),
             bytes(synth.genome(120_000, seed=9)), b"acgtn" * 500]
    for k, w in ((17, 5), (19, 10), (21, 11), (24, 25), (31, 32), (31, 1), (20, 5)):
        batch = [s for s in seqs if len(s) >= k + w - 1]
        res = minimize.minimize_batch(batch, k, w)
        for s, r in zip(batch, res):
            assert np.array_equal(r, oracle.minimize_binary_array(s, k, w)), (k, w, len(s))
    # the single-sequence call (device-wide sort) as well
    g = bytes(synth.genome(300_000, seed=10))
    assert np.array_equal(minimize.minimize_binary(g, 21, 11), oracle.minimize_binary_array(g, 21, 11))
    # k = 32 is beyond the 32 candidate states the machines track: rejected, like w > k + 1
    from sequencealigner_amd import capi
    with pytest.raises(capi.SaError):
        minimize.minimize_batch(seqs[:2], 32, 5)


def test_reads_with_many_repeated_keys_vs_oracle(oracle):
    """The one-kernel sort + unique + output path of read batches keeps a short list of repeated keys and has a
    second way for sequences with more than 64 of them: tiny chunks make every chunk boundary repeat a key."""
    from sequencealigner_amd import capi, minimize, synth
    seqs = [bytes(synth.genome(6000, seed=500 + i)) for i in range(40)] + [b"ACGTTGCAAC" * 500, b"G" * 4000]
    capi.set_option("min_chunk", 8)
    try:
        for k, w in ((15, 5), (6, 7)):
            res = minimize.minimize_batch(seqs, k, w)
            for s, r in zip(seqs, res):
                assert np.array_equal(r, oracle.minimize_binary_array(s, k, w)), (k, w, len(s))
    finally:
        capi.set_option("min_chunk", 0)


def test_single_sequence_bucketed_sort_vs_oracle(oracle):
    """A call on ONE sequence cuts its keys into value-ordered buckets of about a thousand keys and sorts those
    inside LDS.  Buckets are runs of whole fine bins (the top 16 bits of a key), so a value that repeats at
    thousands of positions makes a bucket larger than the LDS sort takes: it is sorted in global memory and
    handed out in slices.  Random sequence, tandem repeats of several periods (a handful of values at tens of
    thousands of positions each), a repeat inside random sequence, tiny sequences, small k (fewer key bits than
    fine-bin bits), k above 16, bytes other than bases (the second run of the exact machine), tiny chunks
    (repeated keys at every chunk boundary)."""
    from sequencealigner_amd import capi, minimize, synth
    g = bytes(synth.genome(300_000, seed=601))
    cases = [
        (g, 15, 5),
        (b"ACGTTGC" * 60_000, 15, 5),
        (b"ACGTTGCAAC" * 30_000 + g[:50_000] + b"AC" * 40_000 + b"T" * 30_000, 15, 5),
        (g[:120_000] + g[1000:31_000] * 6 + g[120_000:200_000], 15, 5),
        (g[:19], 15, 5),
        (g[:700], 15, 5),
        (g, 5, 3),
        (g, 2, 1),
        (b"ACGTTGC" * 60_000, 4, 5),
        (g, 21, 11),
        (g[:100_000] + b"N" * 300 + g[100_000:150_000] + b"n" + g[150_000:], 15, 5),
        (g, 16, 17),
    ]
    for seq, k, w in cases:
        got = minimize.minimize_binary(seq, k, w)
        assert np.array_equal(got, oracle.minimize_binary_array(seq, k, w)), (len(seq), k, w)
    capi.set_option("min_chunk", 8)
    try:
        got = minimize.minimize_binary(g[:90_000], 15, 5)
        assert np.array_equal(got, oracle.minimize_binary_array(g[:90_000], 15, 5))
    finally:
        capi.set_option("min_chunk", 0)


def test_single_sequence_device_wide_sort_vs_oracle(oracle):
    """Sequences beyond 32 M windows are sorted by the device-wide radix sort (the call then synchronises once to
    learn the number of keys); the option no_buckets sends a short sequence the same way."""
    from sequencealigner_amd import capi, minimize, synth
    g = bytes(synth.genome(250_000, seed=611)) + b"ACGTTGCAAC" * 3000 + b"N" * 50 + bytes(synth.genome(9_000, seed=612))
    capi.set_option("no_buckets", 1)
    try:
        for k, w in ((15, 5), (21, 11), (6, 3)):
            assert np.array_equal(minimize.minimize_binary(g, k, w), oracle.minimize_binary_array(g, k, w)), (k, w)
    finally:
        capi.set_option("no_buckets", 0)


def test_inputs_outside_the_supported_range_fail_loudly():
    """What the library does not reproduce is refused, never answered wrongly: MinimizeBinary with w > k + 1 or
    k >= 32, and the decimal Minimize on a sequence with a byte other than ACGT / acgt (the reference drops such
    bytes but keeps counting windows by the original length, src/minimizer.cpp:13-40, 174)."""
    from sequencealigner_amd import capi, minimize
    seq = "ACGTTGCATGCAGTCAGTCGATCGATGCATGCATGACTGACTAGCTAGCTAGTCGATGC"
    for k, w in ((3, 5), (32, 4), (40, 2)):
        with pytest.raises(capi.SaError) as e:
            minimize.minimize_batch([seq], k, w, capi.SA_MIN_BINARY)
        assert e.value.code == capi.SA_ERR_ARGUMENT
    assert len(minimize.minimize_batch([seq], 3, 4, capi.SA_MIN_DECIMAL)[0]) > 0
    for bad in (seq[:20] + "N" + seq[21:], seq + "n", "x" + seq):
        with pytest.raises(capi.SaError) as e:
            minimize.minimize_batch([seq, bad], 3, 4, capi.SA_MIN_DECIMAL)
        assert e.value.code == capi.SA_ERR_ARGUMENT
    # the binary variant takes such bytes (code -1, src/minimizer_binary.cpp:53-78) - see the tests above
    assert len(minimize.minimize_batch([seq[:20] + "N" + seq[21:]], 3, 4, capi.SA_MIN_BINARY)[0]) > 0
