"""GPU parity tests for batched Align(): the HIP path (through the C-ABI) against the golden fixtures
produced by the compiled reference and against the CPU restatement on seeded inputs.  Bit-exact bar:
score, CIGAR string and target_begin must all be identical."""
import ctypes as C
import random

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu

SCHEMES = [(1, -1, -1), (2, -2, -1), (0, -1, -1), (2, -1, -1), (1, -3, -2), (1, 1, -1), (1, -1, 0),
           (1, -1, 1)]


def _run_group(cases):
    """cases: list of dicts sharing (type, match, mismatch, gap). Runs them as ONE ragged batch."""
    from sequencealigner_amd import align
    if not cases:
        return
    c0 = cases[0]
    scores, cigars, tb = align.align_batch([c["q"] for c in cases], [c["t"] for c in cases], c0["type"],
                                           c0["match"], c0["mismatch"], c0["gap"])
    for c, s, cg, b in zip(cases, scores, cigars, tb):
        assert (int(s), cg, int(b)) == (c["score"], c["cigar"], c["target_begin"]), \
            (c["q"], c["t"], c["type"], c["match"], c["mismatch"], c["gap"])


def _grouped(cases):
    groups = {}
    for c in cases:
        groups.setdefault((c["type"], c["match"], c["mismatch"], c["gap"]), []).append(c)
    return groups.values()


def _force_family(name):
    """Pins the kernel family the dispatcher may choose ("lane": the inter-pair lane kernels, packed
    16-bit where exact, else 32-bit; "wave": one pair per wavefront); None restores automatic choice."""
    from sequencealigner_amd import capi
    family, coop, i32, coop_pk, seg = 0, -1, 0, -1, -1
    if name == "cooppkseg":  # as "cooppk", every strip in segments of 64 target rows, a wavefront per segment
        seg = 64
        name = "cooppk"
    if name == "wave32":    # one wave per pair, 32-bit scores only
        i32 = 1
        name = "wave"
    if name == "cooppk":    # as "coop", in the packed form (two pairs per wave) wherever that applies
        coop_pk = 1
        name = "coop"
    if name == "coop":      # wavefront kernels, one wave per (pair, strip) wherever the batch allows it
        family, coop = 2, 1
    elif name == "wave":    # wavefront kernels, one wave per pair
        family, coop = 2, 0
    elif name == "lane":
        family = 1
    capi.set_option("force_family", family)
    capi.set_option("force_coop", coop)
    capi.set_option("force_i32", i32)
    capi.set_option("coop_pk", coop_pk)
    capi.set_option("coop_seg_rows", seg)


@pytest.fixture(params=["auto", "lane", "wave", "wave32", "coop", "cooppk", "cooppkseg"])
def family(request):
    _force_family(None if request.param == "auto" else request.param)
    yield request.param
    _force_family(None)


def test_golden_align_cases(family):
    n = 0
    for g in _grouped(golden_io.load("align_cases.json")):
        _run_group(g)
        n += len(g)
    assert n > 500


def test_golden_benchmark_shapes(family):
    for g in _grouped(golden_io.load("align_shapes.json")):
        _run_group(g)


def test_single_call_mirror():
    from sequencealigner_amd.align import Align, AlignmentType
    # reference unit tests (test/testAlignmentMinimizer.cpp:14,24,34) assert these scores
    assert Align("CTCTGTTCG", 9, "CGTATCTTGA", 10, AlignmentType.Global, 0, -1, -1)[0] == -5
    assert Align("CTCTGAG", 7, "TGTCAGT", 7, AlignmentType.Local, 2, -2, -1)[0] == 6
    assert Align("AGCATGCAAT", 10, "ATCCGAACATCCAATCGAAGC", 21, AlignmentType.Semiglobal, 2, -1, -1)[0] == 14
    assert Align("ACGT", 4, "ACGT", 4, 7, 1, -1, -1)[0] == 9999
    # explicit lengths shorter than the buffers
    assert Align("ACGTTTTT", 4, "TTTACGTAAAA", 7, AlignmentType.Global, 1, -1, -1) == (1, "3D4M", 3)
    assert Align("ACGT", 4, "TTACGT", 6, AlignmentType.Local, 1, -1, -1, want_cigar=False) == (4, None, 2)


def _oracle_cases(oracle, pairs, typ, m, x, g):
    out = []
    for q, t in pairs:
        s, cg, tb = oracle.align(q, t, typ, m, x, g)
        if typ == 1 and s <= 0:
            continue
        if typ == 2 and s <= 0 and g <= 0:
            continue
        out.append({"q": q, "t": t, "type": typ, "match": m, "mismatch": x, "gap": g, "score": s,
                    "cigar": cg, "target_begin": tb})
    return out


def test_ragged_random_vs_oracle(oracle, family):
    rng = random.Random(4242)
    pairs = []
    for _ in range(700):
        alpha = rng.choice(["ACGT", "ACGT", "AC", "ACGTN"])
        T = rng.randint(1, 200)
        t = "".join(rng.choice(alpha) for _ in range(T))
        if rng.random() < 0.6:
            cut = t[rng.randint(0, T // 2):rng.randint(T // 2 + 1, T + 1)]
            q = "".join(c if rng.random() > 0.12 else rng.choice(alpha) for c in cut) or "A"
            if rng.random() < 0.5:
                k = rng.randint(0, len(q))
                q = q[:k] + "".join(rng.choice(alpha) for _ in range(rng.randint(1, 5))) + q[k:]
        else:
            q = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 200)))
        pairs.append((q, t))
    n = 0
    for (m, x, g) in SCHEMES:
        for typ in (0, 1, 2):
            cases = _oracle_cases(oracle, pairs, typ, m, x, g)
            _run_group(cases)
            n += len(cases)
    assert n > 10000


def test_uniform_shapes_vs_oracle(oracle, family):
    from sequencealigner_amd import synth
    q, t = synth.pairs_short_read_window(300, seed=5)
    pairs = [(bytes(a).decode(), bytes(b).decode()) for a, b in zip(q, t)]
    for typ in (0, 1, 2):
        _run_group(_oracle_cases(oracle, pairs, typ, 1, -1, -1))
    q, t = synth.pairs_global_1k(70, seed=6, length=400)
    pairs = [(bytes(a).decode(), bytes(b).decode()) for a, b in zip(q, t)]
    for typ in (0, 1, 2):
        _run_group(_oracle_cases(oracle, pairs, typ, 2, -3, -2))


def test_small_workspace_multiple_passes(oracle):
    """A workspace that holds one block of pairs forces several passes; results must not change."""
    import ctypes as C
    from sequencealigner_amd import align, capi, synth
    q, t = synth.pairs_short_read_window(700, seed=8, read_len=60, window_len=90)
    b = align.DeviceAlignBatch(q, t, 1, 1, -1, -1, workspace_bytes=1)  # clamped to the minimum
    b.run()
    import torch
    torch.cuda.synchronize()
    s, cg, tb = b.scores(), b.cigars(), b.target_begins()
    for i in range(0, 700, 7):
        assert (int(s[i]), cg[i], int(tb[i])) == oracle.align(bytes(q[i]), bytes(t[i]), 1, 1, -1, -1)


def test_leftover_pairs_of_the_last_pass(oracle):
    """900 pairs of 1.1 kbp on the packed wavefront kernel with a workspace for 256 pairs per pass: the 132 pairs
    left over for the last pass are planned as a batch of their own (the cooperative kernel: they are too few
    wavefronts for one wavefront per pair), through offsets into the same arrays.  With the cooperative kernel
    switched off they run as a fourth pass of the packed kernel.  Every pair against the CPU checker."""
    import ctypes as C
    import torch
    from sequencealigner_amd import align, capi, synth
    n = 900
    q, t = synth.pairs_semiglobal_long(n, seed=12, read_len=1100, window_len=1200, lead=40)
    qb, qo = synth.to_blob(q)
    tb_, to = synth.to_blob(t)
    for typ in (2, 0):
        want_s, want_tb = oracle.align_batch(qb, qo, tb_, to, typ, 1, -1, -1, True, 16)
        for coop in (-1, 0):
            capi.set_option("force_coop", coop)
            try:
                probe = align.DeviceAlignBatch(q, t, typ, 1, -1, -1)
                assert probe.kernel_name() == "wave_pk16"
                mn = C.c_size_t(0)
                capi.lib().sa_align_workspace_bytes(C.byref(probe.batch), C.byref(mn))
                per_block = mn.value - 1024          # the minimum is one block of two pairs plus the fixed head
                del probe
                b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, workspace_bytes=1024 + 128 * per_block)
                # the left-over pairs take the cooperative kernel, which synchronises: the query says so, while
                # the same batch over a workspace that holds it in one pass only enqueues work
                assert capi.lib().sa_align_may_synchronize(0, C.byref(b.batch), b.workspace_bytes) == (coop != 0)
                assert capi.lib().sa_align_may_synchronize(0, C.byref(b.batch), 1 << 34) == 0
                b.run()
                torch.cuda.synchronize()
            finally:
                capi.set_option("force_coop", -1)
            assert np.array_equal(b.scores(), want_s) and np.array_equal(b.target_begins(), want_tb)
            cg = b.cigars(list(range(0, n, 37)) + list(range(760, n, 9)))
            for k, i in enumerate(list(range(0, n, 37)) + list(range(760, n, 9))):
                assert cg[k] == oracle.align(bytes(q[i]), bytes(t[i]), typ, 1, -1, -1)[1], (typ, coop, i)


def test_more_pairs_than_one_launch_holds(oracle):
    """4.5 million tiny pairs: the batch exceeds what one pass may launch (2^22 pairs) although the
    workspace would hold it, so the passes are cut by the launch limits."""
    import torch
    from sequencealigner_amd import align, synth
    n = 4_500_000
    rng = np.random.default_rng(77)
    q = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 8))]
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 12))]
    b = align.DeviceAlignBatch(q, t, 0, 1, -1, -1, want_cigar=False)
    assert b.workspace_bytes == b.lib.sa_align_workspace_bytes(C.byref(b.batch), None)
    b.run()
    torch.cuda.synchronize()
    qb, qo = synth.to_blob(q)
    tb, to = synth.to_blob(t)
    s, tbeg = oracle.align_batch(qb, qo, tb, to, 0, 1, -1, -1, True, 16)
    assert np.array_equal(b.scores(), s)
    assert np.array_equal(b.target_begins(), tbeg)


@pytest.mark.parametrize("rare", ["fifth symbol", "fourth symbol", "fifth symbol in a later pass"])
def test_symbol_the_alphabet_sample_misses(oracle, rare):
    """A batch of more than 4 MB has its alphabet sampled, not scanned.  One byte in a word the sample
    skips (word 1 of a blob) is the only occurrence of a symbol: the packing kernel must notice it and the
    batch must still come out exact."""
    import torch
    from sequencealigner_amd import align, synth
    n = 30_000
    q, t = synth.pairs_short_read_window(n, seed=21)
    q, t = np.array(q), np.array(t)
    workspace = None
    if rare == "fifth symbol":
        q[0, 9] = ord("N")
        t[0, 12] = ord("N")
    elif rare == "fifth symbol in a later pass":
        # pair 29 000 starts at query word 543 750, a multiple of the sampling step 3; its byte 9 is in the
        # next word.  A quarter of the full workspace: the pair is met in the last of several passes, after
        # earlier passes have run in the tables form.
        q[29_000, 9] = ord("N")
        probe = align.DeviceAlignBatch(q, t, 1, 1, -1, -1)
        workspace = (probe.lib.sa_align_workspace_bytes(C.byref(probe.batch), None) - 1024) // 4 + 1024
        del probe
    else:  # only A, C, G elsewhere; a single T
        q[q == ord("T")] = ord("G")
        t[t == ord("T")] = ord("G")
        t[0, 12] = ord("T")
    for typ in (1, 0):
        b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, workspace_bytes=workspace)
        assert b.kernel_name() == "lane_pk16"
        b.run()
        torch.cuda.synchronize()
        # head of the workspace: 8 words of byte-presence bits, then "small alphabet", then "violated"
        info = b.d_ws[:40].cpu().numpy().view(np.uint32)
        assert info[8] == 1 and info[9] == 1, "the sample was expected to miss the symbol"
        qb, qo = synth.to_blob(q)
        tb, to = synth.to_blob(t)
        s, tbeg = oracle.align_batch(qb, qo, tb, to, typ, 1, -1, -1, True, 16)
        assert np.array_equal(b.scores(), s)
        assert np.array_equal(b.target_begins(), tbeg)
        for i in (0, 1, n - 1):
            assert b.cigars([i])[0] == oracle.align(bytes(q[i]), bytes(t[i]), typ, 1, -1, -1)[1]


def test_full_size_properties():
    """Full benchmark size (scaled to 200k pairs to bound the time): every CIGAR must re-score to the
    reported score and consume exactly the aligned part of query and target."""
    import torch
    from sequencealigner_amd import align, synth
    n = 200_000
    q, t = synth.pairs_short_read_window(n, seed=9)
    b = align.DeviceAlignBatch(q, t, 1, 1, -1, -1)
    b.run()
    torch.cuda.synchronize()
    scores = b.scores()
    nruns = b.d_nruns[:n].cpu().numpy()
    runs = b.d_runs.cpu().numpy().view(np.uint32).reshape(n, b.stride)
    assert scores.min() > 100 and scores.max() <= 150
    # re-score from runs: M -> +1, X/I/D -> -1
    col = np.arange(b.stride)[None, :]
    valid = col >= (b.stride - nruns)[:, None]
    length = (runs >> 2).astype(np.int64) * valid
    op = runs & 3
    rescored = (length * np.where(op == 0, 1, -1)).sum(axis=1)
    assert np.array_equal(rescored, scores)
    q_used = (length * ((op == 0) | (op == 1) | (op == 2))).sum(axis=1)
    t_used = (length * ((op == 0) | (op == 1) | (op == 3))).sum(axis=1)
    assert q_used.max() <= 150 and t_used.max() <= 300
    tb = b.target_begins().astype(np.int64)
    assert ((tb + t_used) <= 300).all()


@pytest.fixture(params=["wave", "wave32", "coop", "cooppk", "cooppkseg"])
def wave_mode(request):
    _force_family(request.param)
    yield request.param
    _force_family(None)


def test_long_pairs_vs_oracle(oracle, wave_mode):
    """Pairs long enough to need several strips and the in-memory boundary column, with one wave per
    pair and with one wave per (pair, strip)."""
    from sequencealigner_amd import align, synth
    q, t = synth.pairs_semiglobal_long(6, seed=21, read_len=2500, window_len=2800, lead=120)
    pairs = [(bytes(a).decode(), bytes(b).decode()) for a, b in zip(q, t)]
    for typ, (m, x, g) in ((2, (1, -1, -1)), (0, (1, -1, -1)), (1, (2, -3, -2)), (0, (1, -1, 0))):
        cases = _oracle_cases(oracle, pairs, typ, m, x, g)
        assert len(cases) == 6
        b = align.DeviceAlignBatch([c["q"] for c in cases], [c["t"] for c in cases], typ, m, x, g)
        packed = wave_mode == "wave" and typ != 1   # Global / Semiglobal within the 16-bit window
        assert b.kernel_name() == ("wave_i32_coop" if wave_mode == "coop" or (wave_mode.startswith("cooppk") and typ == 1) else
                                   "wave_pk16_coop" if wave_mode.startswith("cooppk") else
                                   "wave_pk16" if packed else "wave_i32")
        _run_group(cases)
    # one pair with unequal lengths, crossing a strip boundary exactly (2048 columns) and not (2049)
    for ql in (2048, 2049, 1025):
        qq, tt = synth.pair_global_long(ql, seed=ql)
        tt = tt[:ql - 37]
        for typ in (0, 1, 2):
            _run_group(_oracle_cases(oracle, [(bytes(qq).decode(), bytes(tt).decode())], typ, 1, -1, -1))


def test_long_gap_runs_vs_oracle(oracle, wave_mode):
    """Queries stitched from two distant pieces of the target, and queries much shorter or longer than the
    target: the paths hold runs of hundreds of D (or I) moves, inside the matrix and along its borders."""
    rng = random.Random(99)
    pairs = []
    for n in range(10):
        T = rng.randint(1500, 2600)
        t = "".join(rng.choice("ACGT") for _ in range(T))
        a, b = rng.randint(0, 200), rng.randint(T - 600, T - 300)
        piece = rng.randint(150, 280)
        q = t[a:a + piece] + t[b:b + piece]           # a gap of ~1000-2000 target rows in the middle
        q = "".join(c if rng.random() > 0.03 else rng.choice("ACGT") for c in q)
        pairs.append((q, t))
        pairs.append((t, q))                           # and the same as a run of insertions
    pairs.append(("ACGTACGTAC", "T" * 700 + "ACGTACGTAC" + "G" * 900))
    pairs.append(("T" * 700 + "ACGTACGTAC" + "G" * 900, "ACGTACGTAC"))
    for typ, (m, x, g) in ((0, (1, -1, -1)), (2, (1, -1, -1)), (1, (2, -3, -1)), (0, (2, -2, 0)), (1, (1, -1, 0))):
        _run_group(_oracle_cases(oracle, pairs, typ, m, x, g))


def test_local_ties_across_strips_vs_oracle(oracle):
    """Local's end cell is the first strict maximum in row-major order (src/alignment.cpp:203-207).  A
    wavefront that walks the strips of its pair one after the other meets the rows of a later strip (larger
    columns) after ALL rows of the earlier one: an equal score in a smaller row of a later strip must win.
    Directed: score 5 ends at (row 50, column 10) and at (row 20, column 1030), both in lane 0 of their
    strips; the alignment must be the one in row 20.  Then many tie-prone pairs over a two-letter alphabet."""
    from sequencealigner_amd import align, synth
    q = bytearray(b"A" * 2048)
    t = bytearray(b"C" * 300)
    q[5:10] = b"GGGGG"
    t[45:50] = b"GGGGG"
    q[1025:1030] = b"TTTTT"
    t[15:20] = b"TTTTT"
    want = oracle.align(bytes(q), bytes(t), 1, 1, -1, -1)
    assert want == (5, "5M", 15)
    for fam in ("wave", "wave32", "coop", "cooppk", "cooppkseg", "lane", None):
        _force_family(fam)
        try:
            s, c, tb = align.align_batch([bytes(q)], [bytes(t)], 1, 1, -1, -1)
            assert (int(s[0]), c[0], int(tb[0])) == want, fam
        finally:
            _force_family(None)
    import torch
    rng = np.random.default_rng(808)
    two = np.frombuffer(b"AC", dtype=np.uint8)
    # lengths beyond the packed wavefront kernel's 11-bit Local window, and (301 pairs: the last wavefront
    # holds one pair) lengths within it for the schemes with match 1
    for lo, hi, n in ((1500, 2500, 300), (1100, 1900, 301)):
        qs = [two[rng.integers(0, 2, size=int(rng.integers(lo, hi)))] for _ in range(n)]
        ts = [two[rng.integers(0, 2, size=int(rng.integers(lo, hi)))] for _ in range(n)]
        qb, qo = synth.to_blob(qs)
        tb_, to = synth.to_blob(ts)
        for (m, x, g) in ((1, -1, -1), (1, -3, -3), (2, -1, -2)):
            # one wavefront per pair (pinned), then the dispatcher's choice for 300 long pairs: the cooperative
            # kernel in passes, whose strips' candidates are merged by the same rule
            for fam, want_kernel in (("wave", "wave_pk16" if hi == 1900 and m == 1 else "wave_i32"),
                                     (None, "wave_i32_coop")):
                _force_family(fam)
                try:
                    b = align.DeviceAlignBatch(qs, ts, 1, m, x, g)
                    assert b.kernel_name() == want_kernel
                    b.run()
                    torch.cuda.synchronize()
                finally:
                    _force_family(None)
                s, tbeg = oracle.align_batch(qb, qo, tb_, to, 1, m, x, g, True, 16)
                assert np.array_equal(b.scores(), s)
                assert np.array_equal(b.target_begins(), tbeg)      # the start row depends on the end cell chosen
                for i in range(0, n, 29):
                    assert b.cigars([i])[0] == oracle.align(bytes(qs[i]), bytes(ts[i]), 1, m, x, g)[1]


def test_cooperative_single_long_pair_vs_oracle(oracle):
    """One 20 kbp x 19 kbp pair: 40 strips in flight at once, each following its left neighbour."""
    from sequencealigner_amd import align, synth
    qq, tt = synth.pair_global_long(20_000, seed=77)
    q, t = bytes(qq).decode(), bytes(tt[:19_000]).decode()
    for typ, (m, x, g) in ((0, (1, -1, -1)), (1, (2, -3, -2)), (2, (1, -1, -1))):
        cases = _oracle_cases(oracle, [(q, t)], typ, m, x, g)
        b = align.DeviceAlignBatch([q], [t], typ, m, x, g)
        assert b.kernel_name() == "wave_i32_coop"
        _run_group(cases)


def _mutated(rng, t, sub, indel):
    out = []
    for c in t:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            out.append(rng.choice("ACGT"))
        out.append(c if rng.random() > sub else rng.choice("ACGT"))
    return "".join(out) or "A"


@pytest.mark.parametrize("cap,packed", [(0, False), (8, False), (0, True), (8, True)])
def test_segmented_traceback_vs_oracle(oracle, cap, packed):
    """The traceback of long Global / Semiglobal pairs by one wavefront per segment of target rows (speculative
    starts, walked again from the column the segment above lands on, joined afterwards): with 64-row segments
    pairs of a few hundred to a few thousand rows take that path.  Related pairs, unrelated ones (the walks meet
    late or not at all inside a segment), two-letter alphabets (ties), long gaps inside the matrix and along its
    borders, targets far longer than the query and the reverse, pairs of unequal length in one batch.  cap = 8:
    every segment runs out of room for its runs, which sends the pair to the one-wavefront walk.  packed: the
    cooperative kernel with two pairs per wavefront, whose direction words the same walk reads."""
    from sequencealigner_amd import align, capi
    rng = random.Random(2024 + cap)
    batches = []
    related = []
    for _ in range(8):
        T = rng.randint(600, 3000)
        t = "".join(rng.choice("ACGT") for _ in range(T))
        related.append((_mutated(rng, t, 0.05, 0.06), t))
    batches.append(related)
    batches.append([("".join(rng.choice("ACGT") for _ in range(rng.randint(500, 2200))),
                     "".join(rng.choice("ACGT") for _ in range(rng.randint(700, 2200)))) for _ in range(6)])
    batches.append([("".join(rng.choice("AC") for _ in range(rng.randint(500, 1500))),
                     "".join(rng.choice("AC") for _ in range(rng.randint(700, 1500)))) for _ in range(6)])
    gaps = []
    for _ in range(4):
        T = rng.randint(1500, 2600)
        t = "".join(rng.choice("ACGT") for _ in range(T))
        a, b = rng.randint(0, 200), rng.randint(T - 600, T - 300)
        piece = rng.randint(150, 280)
        q = _mutated(rng, t[a:a + piece] + t[b:b + piece], 0.03, 0.0)
        gaps.append((q, t))
        gaps.append((t, q))
    gaps.append(("ACGTACGTAC", "T" * 700 + "ACGTACGTAC" + "G" * 900))
    gaps.append(("T" * 700 + "ACGTACGTAC" + "G" * 900, "ACGTACGTAC" * 60))
    batches.append(gaps)
    _force_family("cooppk" if packed else "coop")
    capi.set_option("seg_rows", 64)
    capi.set_option("seg_cap", cap)
    try:
        n = 0
        for pairs in batches:
            for typ, (m, x, g) in ((0, (1, -1, -1)), (2, (1, -1, -1)), (0, (1, -3, -1)), (2, (2, -2, 0)),
                                   (0, (1, 1, -1)), (2, (0, -1, -1))):
                cases = _oracle_cases(oracle, pairs, typ, m, x, g)
                if not cases:
                    continue
                b = align.DeviceAlignBatch([c["q"] for c in cases], [c["t"] for c in cases], typ, m, x, g)
                assert b.kernel_name() == ("wave_pk16_coop" if packed else "wave_i32_coop")
                _run_group(cases)
                n += len(cases)
        assert n > 120
    finally:
        capi.set_option("seg_rows", 0)
        capi.set_option("seg_cap", 0)
        _force_family(None)


def test_cooperative_wait_that_runs_out_fails_the_call(oracle):
    """Every wait inside the cooperative kernels is bounded.  With the bound at one poll (a host-side option;
    the kernels still run to their end) a strip finds its neighbour's rows missing and gives up, the others
    follow, and the call reports it instead of returning a wrong result.  The next call, with the usual bound,
    is unaffected."""
    import torch
    from sequencealigner_amd import align, capi, synth
    qq, tt = synth.pair_global_long(6000, seed=41)
    q, t = bytes(qq).decode(), bytes(tt[:5600]).decode()
    for typ, (m, x, g) in ((0, (1, -1, -1)), (2, (1, -1, -1)), (1, (2, -3, -2))):
        b = align.DeviceAlignBatch([q], [t], typ, m, x, g)
        assert b.kernel_name() == "wave_i32_coop"
        capi.set_option("spin_limit", 1)
        try:
            with pytest.raises(capi.SaError, match="waited too long"):
                b.run()
        finally:
            capi.set_option("spin_limit", 1 << 24)
        torch.cuda.synchronize()
        b.run()
        torch.cuda.synchronize()
        want = oracle.align(q, t, typ, m, x, g)
        assert (int(b.scores()[0]), b.cigars([0])[0], int(b.target_begins()[0])) == want


def test_cooperative_calls_that_only_enqueue(oracle):
    """With the option coop_async a call that runs the cooperative kernels does not wait for them:
    sa_align_may_synchronize answers 0, the results are the usual ones, and a wait that runs out (the bound at one
    poll, a host-side option; the kernels still run to their end) shows as the marker score in every pair of the
    batch instead of a failed call."""
    import ctypes as C
    import torch
    from sequencealigner_amd import align, capi, synth
    rng = np.random.default_rng(12)
    qs, ts = [], []
    for p in range(6):
        t = synth.random_bases(rng, 5200 - 100 * p)
        qs.append(bytes(synth.fit_length(rng, synth.edit(rng, t[:4800], 0.04, 0.03, 0.03), 4800 - 50 * p)))
        ts.append(bytes(t))
    capi.set_option("coop_async", 1)
    try:
        for typ in (0, 2):
            b = align.DeviceAlignBatch(qs, ts, typ, 1, -1, -1)
            assert b.kernel_name().endswith("_coop")
            assert capi.lib().sa_align_may_synchronize(0, C.byref(b.batch), b.workspace_bytes) == 0
            capi.set_option("spin_limit", 1)
            try:
                b.run()   # returns without an error
            finally:
                capi.set_option("spin_limit", 1 << 24)
            torch.cuda.synchronize()
            assert (b.scores() == -2**31).all()
            b.run()
            torch.cuda.synchronize()
            for i in range(6):
                want = oracle.align(qs[i], ts[i], typ, 1, -1, -1)
                assert (int(b.scores()[i]), b.cigars([i])[0], int(b.target_begins()[i])) == want, (typ, i)
        # the packed cooperative kernel: 128 pairs, scores against the linear-memory scorer
        t = synth.random_bases(rng, (128, 4400))
        q = np.stack([synth.fit_length(rng, synth.edit(rng, t[p, :4300], 0.04, 0.03, 0.03), 4300) for p in range(128)])
        b = align.DeviceAlignBatch(q, t, 2, 1, -1, -1)
        assert b.kernel_name() == "wave_pk16_coop"
        assert capi.lib().sa_align_may_synchronize(0, C.byref(b.batch), b.workspace_bytes) == 0
        capi.set_option("spin_limit", 1)
        try:
            b.run()
        finally:
            capi.set_option("spin_limit", 1 << 24)
        torch.cuda.synchronize()
        assert (b.scores() == -2**31).all()
        b.run()
        torch.cuda.synchronize()
        qb, qo = synth.to_blob(q)
        tb, to = synth.to_blob(t)
        assert np.array_equal(b.scores(), oracle.score_linear_batch(qb, qo, tb, to, 2, 1, -1, -1, 8))
    finally:
        capi.set_option("coop_async", 0)


def test_empty_and_odd_batches_vs_oracle(oracle, family):
    """Empty queries / targets next to ordinary pairs, and batches of 1 and 3 pairs (a packed wavefront or
    lane then holds a pair without a partner); the meaning of the empty cases is the one the CPU
    restatement fixes (the reference itself reads out of bounds there)."""
    from sequencealigner_amd import align
    rng = random.Random(17)
    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    base = [("", "ACGT"), ("ACGT", ""), ("", ""), (rnd(70), rnd(90)), ("", rnd(300)), (rnd(300), ""),
            (rnd(1100), rnd(1300)), (rnd(5), rnd(2000)), (rnd(2100), rnd(7))]
    for pairs in (base, base[3:4], base[:3], base[5:8]):
        for typ in (0, 1, 2):
            want = [oracle.align(q, t, typ, 1, -1, -1) for q, t in pairs]
            scores, cigars, tb = align.align_batch([q for q, _ in pairs], [t for _, t in pairs], typ, 1, -1, -1)
            got = [(int(s_), c_, int(b_)) for s_, c_, b_ in zip(scores, cigars, tb)]
            assert got == want, (typ, family, [(len(q), len(t)) for q, t in pairs])


def test_cigar_stats_on_device_vs_text(oracle):
    """Block length and per-op lengths from the run arrays equal what the mapper's text walk gives
    (block length) and a plain count over the CIGAR text (per op)."""
    import re
    import torch
    from sequencealigner_amd import align, synth
    q, t = synth.pairs_short_read_window(3000, seed=41)
    for typ in (0, 1, 2):
        b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1)
        b.run()
        torch.cuda.synchronize()
        block, ops = b.cigar_stats()
        cigars = b.cigars()
        for p in range(0, 3000, 13):
            abl, ml = oracle.cigar_block_match(cigars[p])
            assert (int(block[p]), ml) == (abl, 0)
            want = [sum(int(n) for n, o in re.findall(r"(\d+)([MXID])", cigars[p]) if o == op) for op in "MXID"]
            assert list(map(int, ops[p])) == want, cigars[p]


def test_arbitrary_byte_alphabets_vs_oracle(oracle, family):
    """Align() compares symbols as opaque bytes (src/alignment.cpp:63,176,281): protein-like alphabets and
    raw bytes, NUL and values above 127 included, must behave like letters."""
    rng = random.Random(77)
    protein = b"ACDEFGHIKLMNPQRSTVWY"
    anybyte = bytes(range(256))
    pairs = []
    for alpha in (protein, anybyte, b"\x00\xff\x80A"):
        for _ in range(120):
            T = rng.randint(1, 160)
            t = bytes(rng.choice(alpha) for _ in range(T))
            cut = t[rng.randint(0, T // 2):rng.randint(T // 2 + 1, T + 1)]
            q = bytes(c if rng.random() > 0.15 else rng.choice(alpha) for c in cut) or alpha[:1]
            pairs.append((q, t))
    for (m, x, g) in ((1, -1, -1), (2, -3, -2)):
        for typ in (0, 1, 2):
            _run_group(_oracle_cases(oracle, pairs, typ, m, x, g))


def test_packed_wavefront_kernel_tables_per_wavefront(oracle):
    """The packed wavefront kernel looks up substitution scores in tables when the sequences of a wavefront's two
    pairs hold at most four distinct byte values, and compares symbols otherwise; the choice is made per wavefront.
    One batch mixes: plain ACGT pairs, pairs with a fifth symbol in one sequence only (the partner pair of the
    same wavefront then compares too), four-symbol sets of other bytes, two-symbol pairs; an odd number of pairs."""
    import torch
    from sequencealigner_amd import align, synth
    rng = np.random.default_rng(4242)

    def pair(alphabet, L):
        t = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), size=L + int(rng.integers(0, 80)))]
        q = t[:L].copy()
        flip = rng.random(L) < 0.08
        q[flip] = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), size=int(flip.sum()))]
        return q, t

    qs, ts = [], []
    for i in range(41):
        L = int(rng.integers(600, 1500))
        kind = i % 5
        q, t = pair([b"ACGT", b"ACGT", b"acgt", b"\x00\x01\xfe\xff", b"AC"][kind], L)
        if kind == 1:      # a fifth symbol, in the target only
            t = t.copy()
            t[int(rng.integers(0, len(t)))] = ord("N")
        qs.append(q)
        ts.append(t)
    qb, qo = synth.to_blob(qs)
    tb_, to = synth.to_blob(ts)
    for typ, (m, x, g) in ((0, (1, -1, -1)), (2, (2, -1, -2)), (1, (1, -1, -1)), (0, (2, -3, -2)), (2, (3, -1, -2))):
        _force_family("wave")     # not the cooperative kernel, which a batch this small would get
        try:
            b = align.DeviceAlignBatch(qs, ts, typ, m, x, g)
            assert b.kernel_name() == "wave_pk16", (typ, m, x, g)
            b.run()
            torch.cuda.synchronize()
        finally:
            _force_family(None)
        s, tbeg = oracle.align_batch(qb, qo, tb_, to, typ, m, x, g, True, 16)
        ok = np.ones(len(qs), bool) if typ == 0 else s > 0
        assert np.array_equal(b.scores()[ok], s[ok]), (typ, m, x, g)
        assert np.array_equal(b.target_begins()[ok], tbeg[ok]), (typ, m, x, g)
        for i in range(0, len(qs), 3):
            if ok[i]:
                assert b.cigars([i])[0] == oracle.align(bytes(qs[i]), bytes(ts[i]), typ, m, x, g)[1], (typ, i)


def test_small_workspace_sends_a_large_batch_to_the_packed_cooperative_kernel(oracle):
    """2100 Semiglobal pairs of 1500 x 300 plan as one wavefront per two pairs; over a workspace that holds only a
    hundred wavefronts' worth of direction words at a time that kernel would leave nine SIMDs in ten idle in every
    pass, so the call takes the packed cooperative kernel instead (and says so: it synchronises).  Same scores
    as the CPU checker's linear-memory scorer, CIGARs of a sample verbatim."""
    import ctypes as C
    import torch
    from sequencealigner_amd import align, capi, synth
    n = 2100
    rng = np.random.default_rng(77)
    t = synth.random_bases(rng, (n, 300))
    q = np.empty((n, 1500), np.uint8)
    for p in range(n):
        q[p] = synth.fit_length(rng, synth.edit(rng, np.resize(t[p], 1500), 0.05, 0.02, 0.02), 1500)
    probe = align.DeviceAlignBatch(q, t, 2, 1, -1, -1)
    assert probe.kernel_name() == "wave_pk16"
    mn = C.c_size_t(0)
    capi.lib().sa_align_workspace_bytes(C.byref(probe.batch), C.byref(mn))
    assert capi.lib().sa_align_may_synchronize(0, C.byref(probe.batch), probe.workspace_bytes) == 0
    del probe
    b = align.DeviceAlignBatch(q, t, 2, 1, -1, -1, workspace_bytes=100 * mn.value)
    assert capi.lib().sa_align_may_synchronize(0, C.byref(b.batch), b.workspace_bytes) == 1
    b.run()
    torch.cuda.synchronize()
    qb, qo = synth.to_blob(q)
    tb, to = synth.to_blob(t)
    assert np.array_equal(b.scores(), oracle.score_linear_batch(qb, qo, tb, to, 2, 1, -1, -1, 8))
    for i in range(0, n, 211):
        want = oracle.align(bytes(q[i]), bytes(t[i]), 2, 1, -1, -1)
        assert (int(b.scores()[i]), b.cigars([i])[0], int(b.target_begins()[i])) == want, i


def test_a_launch_with_more_strips_than_wavefronts_runs_in_segments(oracle):
    """3400 Semiglobal pairs of 2100 x 1400 (ragged): 1700 units of three strips each, more strips than the GPU holds
    wavefronts of the packed cooperative kernel, so the dispatcher's own choice is that kernel with every strip cut
    into segments of 1024 target rows (two per strip here, the second one short).  Every score equals the CPU
    checker's linear-memory scorer, a sample of pairs its full result verbatim."""
    import torch
    from sequencealigner_amd import align, synth
    n = 3400
    rng = np.random.default_rng(2100)
    qs, ts = [], []
    for p in range(n):
        t = synth.random_bases(rng, 1400 - (p % 7) * 13)
        src = np.resize(t, 2100)
        q = synth.fit_length(rng, synth.edit(rng, src, 0.04, 0.03, 0.03), 2100 - (p % 5) * 31)
        qs.append(bytes(q))
        ts.append(bytes(t))
    b = align.DeviceAlignBatch(qs, ts, 2, 1, -1, -1)
    assert b.kernel_name() == "wave_pk16_coop"
    b.run()
    torch.cuda.synchronize()
    qb, qo = synth.to_blob(qs)
    tb, to = synth.to_blob(ts)
    assert np.array_equal(b.scores(), oracle.score_linear_batch(qb, qo, tb, to, 2, 1, -1, -1, 8))
    for i in range(0, n, 487):
        want = oracle.align(qs[i], ts[i], 2, 1, -1, -1)
        assert (int(b.scores()[i]), b.cigars([i])[0], int(b.target_begins()[i])) == want, i
