"""CPU suite: host-side logic that needs no GPU (input generators, CIGAR text, shard planning)."""
import numpy as np

from sequencealigner_amd import dist, synth


def test_generators_are_deterministic_and_shaped():
    q1, t1 = synth.pairs_short_read_window(50, seed=3)
    q2, t2 = synth.pairs_short_read_window(50, seed=3)
    assert np.array_equal(q1, q2) and np.array_equal(t1, t2)
    assert q1.shape == (50, 150) and t1.shape == (50, 300)
    assert set(np.unique(t1)) <= set(b"ACGT")
    q, t = synth.pairs_global_1k(3, seed=1)
    assert q.shape == t.shape == (3, 1000)
    ident = (q == t).mean()
    assert 0.3 < ident < 1.0
    q, t = synth.pairs_semiglobal_long(2, read_len=2000, window_len=2200, lead=100)
    assert q.shape == (2, 2000) and t.shape == (2, 2200)
    blob, off = synth.to_blob([b"ACG", b"", b"TT"])
    assert bytes(blob) == b"ACGTT" and off.tolist() == [0, 3, 3, 5]


def test_cigar_text():
    import pytest
    torch = pytest.importorskip("torch")  # align.py imports torch for device memory
    from sequencealigner_amd.align import runs_to_cigar
    assert runs_to_cigar([(3 << 2) | 3, (4 << 2) | 0, (1 << 2) | 1, (12 << 2) | 2]) == "3D4M1X12I"
    assert runs_to_cigar([]) == ""


def test_shard_by_cost_balances_and_partitions():
    rng = np.random.default_rng(0)
    costs = rng.integers(1, 10_000, size=5000) * rng.integers(1, 10_000, size=5000)
    for k in (1, 2, 3, 8):
        shards = dist.shard_by_cost(costs, k)
        allidx = np.concatenate(shards)
        assert np.array_equal(np.sort(allidx), np.arange(5000))
        loads = np.array([costs[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.05
    shards = dist.shard_by_cost(np.full(1000, 45000), 8)
    assert [len(s) for s in shards] == [125] * 8


def test_cigar_block_and_match_length_restatement(oracle):
    """getAlignmentBlockLengthAndMatchLength (reference src/mapper.cpp:371-399), hand-traced: the block
    length sums every count; the match length only counts a lower-case 'm', which CompressCigar never
    emits, so it is 0 on real CIGARs."""
    assert oracle.cigar_block_match("3M2X1I") == (6, 0)
    assert oracle.cigar_block_match("18D2M12I") == (32, 0)
    assert oracle.cigar_block_match("10m5X3m") == (18, 13)
    assert oracle.cigar_block_match("") == (0, 0)
    assert oracle.cigar_block_match("150M") == (150, 0)


def test_paf_line_layout():
    """generatePAFString (reference src/mapper.cpp:354-369), hand-expanded."""
    from sequencealigner_amd import paf
    assert paf.paf_line("read1", 150, 3, 148, "ref", 4600000, 1000, 1147, 0, 149, "145M4X") == \
        "read1\t150\t3\t148\t + \tref\t4600000\t1000\t1147\t0\t149\t0\t145M4X\n"
    assert paf.paf_line("r", 10, -1, -1, "t", 20, -1, -1, 0, 0) == "r\t10\t-1\t-1\t + \tt\t20\t-1\t-1\t0\t0\t0\n"
    lines = paf.paf_lines(["a", "b"], [5, 6], [(0, 5, 10, 15), (1, 6, 20, 26)], "t", 100, [5, 7], ["5M", "3M2D2M"])
    assert lines[1] == "b\t6\t1\t6\t + \tt\t100\t20\t26\t0\t7\t0\t3M2D2M\n"


def test_chain_restatement_hand_cases(oracle):
    """Seed chaining (reference src/mapper.cpp:172-352), small hand-traced cases."""
    import numpy as np
    ref = np.array([[5, 100, 1], [7, 10, 1], [7, 50, 0], [7, 90, 1], [9, 200, 1]], dtype=np.uint32)
    reads = np.array([[5, 30, 1], [7, 3, 1], [9, 60, 1],      # read 0
                      [8, 1, 1],                               # read 1: no common value
                      [7, 4, 0]], dtype=np.uint32)             # read 2: one match on the other strand
    off = np.array([0, 3, 4, 5], dtype=np.uint64)
    out, nm = oracle.chain_batch(ref, reads, off, 0.0)
    # read 0: matches sorted by (read pos, ref pos): (3,10) (3,90) (30,100) (60,200); only the reference
    # positions must increase, so all four chain
    assert list(nm) == [4, 0, 1]
    assert out[0].tolist() == [4, 10, 200, 3, 60]
    assert out[1].tolist() == [0, 0, 0, 0, 0]
    assert out[2].tolist() == [1, 50, 50, 4, 4]
    # f = 0.4: (int)(3 * 0.4) = 1 value dropped: the one with the most occurrences (7)
    out, nm = oracle.chain_batch(ref, reads, off, 0.4)
    assert list(nm) == [2, 0, 0] and out[0].tolist() == [2, 100, 200, 30, 60]


def test_sequence_file_rules_of_the_restatement(oracle):
    """Hand cases for the FASTA / FASTQ rules the device parser is checked against (the parser behind the
    reference's mapper is an un-vendored submodule: parity unpinned)."""
    seq, off, ns, nl, qs = oracle.parse(b">r1 desc\nACGT\nAC\n>r2\nGG\r\nTT\n", 0)
    assert bytes(seq) == b"ACGTACGGTT" and off.tolist() == [0, 6, 10]
    assert ns.tolist() == [1, 18] and nl.tolist() == [2, 2]
    # a '>' inside a header line is text; quality lines may start with '@' or '+'
    seq, off, ns, nl, qs = oracle.parse(b">a>b x\nAC\n", 0)
    assert bytes(seq) == b"AC" and nl.tolist() == [3]
    seq, off, ns, nl, qs = oracle.parse(b"@q1 x\nACGT\n+\n@III\n@q2\nAC\n+q2\n+I", 1)
    assert bytes(seq) == b"ACGTAC" and off.tolist() == [0, 4, 6] and qs.tolist() == [13, 29]
    for bad, fmt in ((b"ACGT\n", 0), (b">\nAC\n", 0), (b">a\n", 0), (b"@a\nAC\n+\nI\n", 1), (b"@a\nAC\n+\n", 1)):
        assert oracle.parse(bad, fmt) is None
    assert oracle.parse(b"", 0)[1].tolist() == [0]


def test_file_name_tests_split_and_stats():
    from sequencealigner_amd import ingest
    # the mapper tests ".fa" first, as a substring: a ".fastq" name is taken for FASTA (src/mapper.cpp:135-150)
    assert ingest.format_of("ref.fasta") == ingest.FASTA and ingest.format_of("reads.fastq") == ingest.FASTA
    assert ingest.format_of("reads.fq") == ingest.FASTQ and ingest.format_of("reads.sanfastq") == ingest.FASTQ
    assert ingest.format_of("reads.txt") is None
    short, long_ = ingest.split_fragments([10, 5000, 4999, 70000])
    assert short.tolist() == [0, 2] and long_.tolist() == [1, 3]
    st = ingest.stats([300, 100, 500, 200], ["a", "b", "c", "d"])
    # N50 as the mapper computes it: file order, first fragment at which the running sum exceeds half
    assert st == {"count": 4, "total": 1100, "longest": 500, "longest_name": "c", "shortest": 100,
                  "shortest_name": "b", "average": 275.0, "n50": 500}


def test_failed_host_call_hands_nothing_out():
    """A host batch call that fails leaves NULL in every result slot (nothing for the caller to free) and
    reports through its return code; checked on calls that fail before any GPU work."""
    import ctypes as C
    from sequencealigner_amd import capi
    lib = capi.lib()
    n = 3
    seqs = (C.c_char_p * n)(b"ACGT", b"ACGTACGT", b"AC")
    lens = (C.c_uint32 * n)(4, 8, 2)
    scores = (C.c_int32 * n)()
    tb = (C.c_uint32 * n)()
    cig = (C.c_void_p * n)(0xDEAD0, 0xDEAD0, 0xDEAD0)   # stale values a careless caller would free
    rc = lib.sa_align_batch_host(seqs, lens, seqs, lens, n, 7, 1, -1, -1, 0, scores, tb, cig)
    assert rc == capi.SA_ERR_ARGUMENT and b"alignment type" in lib.sa_last_error()
    assert [cig[i] for i in range(n)] == [None] * n
    out = (C.c_void_p * n)(0xDEAD0, 0xDEAD0, 0xDEAD0)
    counts = (C.c_uint64 * n)()
    rc = lib.sa_minimize_batch_host(None, lens, n, 15, 5, 0, 0, out, counts)
    assert rc == capi.SA_ERR_ARGUMENT
    assert [out[i] for i in range(n)] == [None] * n


def _shard_units(cost, n_shards, block):
    import ctypes as C
    from sequencealigner_amd import capi
    cost = np.ascontiguousarray(cost, dtype=np.uint64)
    shard_of = np.full(cost.size, 0xFFFFFFFF, dtype=np.uint32)
    order = np.full(cost.size, 0xFFFFFFFF, dtype=np.uint32)
    capi.check(capi.lib().sa_shard_units(cost.ctypes.data, cost.size, n_shards, block, shard_of.ctypes.data,
                                         order.ctypes.data))
    return shard_of, order


def test_device_batch_queue_dealing_rule():
    """sa_shard_units is the rule by which sa_align_batch_host / sa_minimize_batch_host split a batch over the
    GPUs (the replacement of the reference's per-read Submit loop, src/mapper.cpp:665-675): a partition of the
    units, balanced by cost, every shard in decreasing cost order; equal costs keep the input order."""
    rng = np.random.default_rng(3)
    cost = (rng.integers(50, 3000, size=20_000) * rng.integers(50, 3000, size=20_000)).astype(np.uint64)
    for n_shards, block in ((1, 16), (2, 16), (4, 1024), (8, 16), (8, 1)):
        shard_of, order = _shard_units(cost, n_shards, block)
        assert shard_of.max() == n_shards - 1
        assert np.array_equal(np.sort(order), np.arange(cost.size))          # a partition, every unit once
        loads = np.bincount(shard_of, weights=cost.astype(np.float64), minlength=n_shards)
        assert loads.max() / loads.mean() < (1.01 if block <= 16 else 1.06)
        at = 0
        for d in range(n_shards):                                            # order lists shard after shard
            members = order[at:at + int((shard_of == d).sum())]
            at += members.size
            assert (shard_of[members] == d).all()
            assert (np.diff(cost[members].astype(np.int64)) <= 0).all()      # decreasing cost inside a shard
    # equal costs: no sort, blocks of `block` consecutive units, round-robin
    shard_of, order = _shard_units(np.full(10, 7), 3, 2)
    assert shard_of.tolist() == [0, 0, 1, 1, 2, 2, 0, 0, 1, 1]
    assert order.tolist() == [0, 1, 6, 7, 2, 3, 8, 9, 4, 5]
    # fewer units than shards, no units, bad arguments
    assert _shard_units([5, 9], 4, 16)[0].tolist() == [0, 0]
    assert _shard_units([], 4, 16)[0].size == 0
    from sequencealigner_amd import capi
    assert capi.lib().sa_shard_units(None, 0, 0, 16, None, None) == capi.SA_ERR_ARGUMENT


def test_options_are_set_by_name_only():
    from sequencealigner_amd import capi
    lib = capi.lib()
    assert lib.sa_set_option(b"no_such_option", 1) == capi.SA_ERR_ARGUMENT
    assert b"unknown option" in lib.sa_last_error()
    import ctypes
    b = capi.AlignBatch()
    b.n, b.max_qlen, b.max_tlen, b.min_qlen, b.min_tlen = 1_000_000, 150, 300, 150, 300
    b.type, b.match, b.mismatch, b.gap = 1, 1, -1, -1
    try:
        capi.set_option("force_i32", 1)
        assert lib.sa_align_kernel_name(ctypes.byref(b)) == b"lane_i32"
        capi.set_option("force_family", 2)
        assert lib.sa_align_kernel_name(ctypes.byref(b)) == b"wave_i32"
    finally:
        capi.set_option("force_i32", 0)
        capi.set_option("force_family", 0)
    assert lib.sa_align_kernel_name(ctypes.byref(b)) == b"lane_pk16"


def test_command_line_surface_of_the_mapper():
    """The option string and defaults of the mapper's main() (reference src/mapper.cpp:446-529, :467-471), its
    index figures (:198-222) and report layouts, without touching the GPU."""
    from sequencealigner_amd import cli
    opt, rest = cli.parse_args([])
    assert (opt["alignment_type"], opt["match"], opt["mismatch"], opt["gap"], opt["cigar"], opt["k"], opt["w"],
            opt["threads"], opt["f"]) == (0, 1, -1, 0, False, 15, 5, 5, 0.001) and rest == []
    # short -c takes an argument (option string "c:"), --cigar takes none; options may follow the file names
    opt, rest = cli.parse_args(["-a", "2", "ref.fa", "-c", "reads.fq", "reads2.fq", "--gap=-2", "-k", "19x", "-n", "abc"])
    assert rest == ["ref.fa", "reads2.fq"] and opt["cigar"] and opt["gap"] == -2
    assert opt["k"] == 19 and opt["mismatch"] == 0      # atoi: leading digits, 0 without any
    opt, rest = cli.parse_args(["--cigar", "a.fa", "b.fq", "--thread_num", "12", "--frequent", "0.01"])
    assert opt["cigar"] and rest == ["a.fa", "b.fq"] and opt["threads"] == 12 and opt["f"] == 0.01
    # (occurrences, value) pairs in increasing order: (1,5) (1,11) (2,9) (3,7); f = 0.3 -> index 1 from the top
    assert cli.index_figures([5, 7, 7, 7, 9, 9, 11], 0.3) == (4, 2, 2)
    assert cli.index_figures([5, 7, 7, 7, 9, 9, 11], 0.0) == (4, 2, 3)
    lines = cli.help_text().splitlines()
    assert len(lines) == 19 and all(len(x) == len(lines[0]) for x in lines)
    assert lines[1].split("|")[1:5] == ["   command", " " * 19 + "explanation", "parameter ", "      default values"]
    text = cli.stats_text({"count": 4, "total": 1100, "longest": 500, "longest_name": "c", "shortest": 100,
                           "shortest_name": "b", "average": 275.0, "n50": 500})
    assert "Average length: 275\n" in text and text.startswith("---------------Statistics---------------\n")
    assert "Singleton Fraction of refence genome: 0.5\n" in cli.index_text(4, 2, 3)
    assert "alignment score: 7\ntaget begin: 3\n" in cli.alignment_text("q", 10, "t", 12, 0, 1, -1, 0, 3, 7)
