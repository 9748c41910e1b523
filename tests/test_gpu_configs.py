"""GPU tests at the sizes of the benchmark configurations that the small parity tests do not reach:
Smith-Waterman 10^6 pairs of 150 x 300 (the whole headline batch), semi-global 10 kbp x 11 kbp pairs on the
packed wavefront kernel, and one global 100 kbp x 100 kbp pair on the cooperative kernel.  Where the
8-byte-per-cell CPU matrix is affordable the results are compared with the CPU checker outright; beyond that
every CIGAR must re-score to the reported score, consume exactly the aligned spans, put M on equal and X on
unequal symbols, and the score must equal the CPU checker's linear-memory scorer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rescore(runs, nruns, stride, match, mismatch, gap):
    """Per pair: (score of the path, query symbols consumed, target symbols consumed) from the run slots."""
    col = np.arange(stride, dtype=np.int64)[None, :]
    valid = col >= (stride - nruns.astype(np.int64))[:, None]
    length = (runs >> 2).astype(np.int64) * valid
    op = runs & 3
    unit = np.where(op == 0, match, np.where(op == 1, mismatch, gap))
    return ((length * unit).sum(axis=1), (length * (op != 3)).sum(axis=1), (length * (op != 2)).sum(axis=1))


def _walk(runs, q, t, qi, ti):
    """Follows one pair's runs from (qi, ti): M runs must cover equal symbols, X runs unequal ones.
    -> (qi, ti) at the end of the path."""
    q, t = np.frombuffer(q, np.uint8), np.frombuffer(t, np.uint8)
    for r in runs.tolist():
        n, op = r >> 2, r & 3
        if op == 0:
            assert np.array_equal(q[qi:qi + n], t[ti:ti + n]), ("M over unequal symbols", qi, ti, n)
        elif op == 1:
            assert n > 0 and not (q[qi:qi + n] == t[ti:ti + n]).any(), ("X over equal symbols", qi, ti, n)
        if op != 3:
            qi += n
        if op != 2:
            ti += n
    return qi, ti


def test_config2_local_full_batch_vs_oracle(oracle):
    """All 10^6 pairs of the headline workload: score and target_begin of every pair equal the CPU checker's,
    every CIGAR re-scores to its score and stays inside the read and the window, a sample of CIGAR strings is
    compared verbatim."""
    import torch
    from sequencealigner_amd import align, synth
    n = 1_000_000
    q, t = synth.pairs_short_read_window(n, seed=1000)
    b = align.DeviceAlignBatch(q, t, 1, 1, -1, -1)
    assert b.kernel_name() == "lane_pk16"
    b.run()
    torch.cuda.synchronize()
    scores, tbeg = b.scores(), b.target_begins().astype(np.int64)
    qb, qo = synth.to_blob(q)
    tb, to = synth.to_blob(t)
    want_s, want_tb = oracle.align_batch(qb, qo, tb, to, 1, 1, -1, -1, True, 16)
    assert np.array_equal(scores, want_s)
    assert np.array_equal(tbeg, want_tb)
    nruns = b.d_nruns[:n].cpu().numpy()
    for lo in range(0, n, 100_000):      # in slices: the run slots of the whole batch are 1.8 GB
        hi = lo + 100_000
        runs = b.d_runs[lo * b.stride:hi * b.stride].cpu().numpy().view(np.uint32).reshape(hi - lo, b.stride)
        rs, qu, tu = _rescore(runs, nruns[lo:hi], b.stride, 1, -1, -1)
        assert np.array_equal(rs, scores[lo:hi])
        assert qu.max() <= 150 and (tbeg[lo:hi] + tu <= 300).all()
    for i in range(0, n, 9973):
        assert b.cigars([i])[0] == oracle.align(bytes(q[i]), bytes(t[i]), 1, 1, -1, -1)[1], i


def test_config3_semiglobal_10k_x_11k_packed_wavefront_kernel(oracle):
    """512 resident pairs of 10 kbp reads x 11 kbp windows through wave_pk16 (the kernel the 200 000-pair
    workload runs on): the first 3 pairs equal the CPU checker's full-matrix result (880 MB each) verbatim,
    every pair's score equals the linear-memory scorer, every CIGAR re-scores and consumes the whole read."""
    import torch
    from sequencealigner_amd import align, synth
    n = 512
    q, t = synth.pairs_semiglobal_long(n, seed=3)
    assert q.shape == (n, 10_000) and t.shape == (n, 11_000)
    from sequencealigner_amd import capi
    capi.set_option("force_coop", 0)   # (512 pairs alone would take the cooperative kernel, see below)
    try:
        b = align.DeviceAlignBatch(q, t, 2, 1, -1, -1)
        assert b.kernel_name() == "wave_pk16"
        b.run()
        torch.cuda.synchronize()
    finally:
        capi.set_option("force_coop", -1)
    scores = b.scores()
    qb, qo = synth.to_blob(q)
    tb, to = synth.to_blob(t)
    assert np.array_equal(scores, oracle.score_linear_batch(qb, qo, tb, to, 2, 1, -1, -1, 16))
    assert scores.min() > 5000            # the reads really align (about 10 % edits)
    nruns = b.d_nruns[:n].cpu().numpy()
    runs = b.d_runs.cpu().numpy().view(np.uint32).reshape(n, b.stride)
    rs, qu, tu = _rescore(runs, nruns, b.stride, 1, -1, -1)
    assert np.array_equal(rs, scores)
    assert (qu == 10_000).all() and (tu <= 11_000).all() and (tu > 9_000).all()
    assert (b.target_begins() == 1).all()  # a Semiglobal CIGAR never starts with D (src/alignment.cpp:129-136)
    cig = b.cigars([0, 1, 2])
    for i in range(3):
        want = oracle.align(bytes(q[i]), bytes(t[i]), 2, 1, -1, -1)
        assert (int(scores[i]), cig[i], 1) == want, i
    # Left to the dispatcher, 512 long pairs are too few wavefronts for the one-wavefront-per-two-pairs kernel: they
    # run on its cooperative form (a wavefront per strip) and give the same runs.
    bc = align.DeviceAlignBatch(q, t, 2, 1, -1, -1)
    assert bc.kernel_name() == "wave_pk16_coop"
    bc.run()
    torch.cuda.synchronize()
    assert np.array_equal(bc.scores(), scores) and np.array_equal(bc.target_begins(), b.target_begins())
    nruns_c = bc.d_nruns[:n].cpu().numpy()
    assert np.array_equal(nruns_c, nruns)
    runs_c = bc.d_runs.cpu().numpy().view(np.uint32).reshape(n, bc.stride)
    for i in range(n):
        assert np.array_equal(runs_c[i, bc.stride - nruns[i]:], runs[i, b.stride - nruns[i]:]), i
    del bc, runs_c
    # the same 3 pairs as a batch of their own, pinned to the packed kernel (so few pairs would otherwise run
    # on the cooperative kernel)
    capi.set_option("force_coop", 0)
    try:
        b3 = align.DeviceAlignBatch(q[:3], t[:3], 2, 1, -1, -1)
        assert b3.kernel_name() == "wave_pk16"
        b3.run()
        torch.cuda.synchronize()
        assert b3.cigars() == cig and np.array_equal(b3.scores(), scores[:3])
    finally:
        capi.set_option("force_coop", -1)


def test_config5_global_100k_x_100k_single_pair(oracle):
    """One 100 kbp x 100 kbp pair (10^10 cells, 2.5 GB of direction bits) on the cooperative kernel: the score
    equals the CPU checker's linear-memory scorer (the full matrix would need 80 GB), the CIGAR re-scores to
    it, consumes exactly query and target, and has M on equal and X on unequal symbols."""
    import torch
    from sequencealigner_amd import align, synth
    qq, tt = synth.pair_global_long(100_000, seed=5)
    q, t = bytes(qq), bytes(tt)
    b = align.DeviceAlignBatch([q], [t], 0, 1, -1, -1)
    assert b.kernel_name() == "wave_i32_coop"
    b.run()
    torch.cuda.synchronize()
    score = int(b.scores()[0])
    nr = int(b.d_nruns[0].item())
    runs = b.d_runs.cpu().numpy().view(np.uint32)[b.stride - nr:b.stride]
    rs, qu, tu = _rescore(runs[None, :], np.array([nr]), nr, 1, -1, -1)
    assert (int(rs[0]), int(qu[0]), int(tu[0])) == (score, 100_000, 100_000)
    assert _walk(runs, q, t, 0, 0) == (100_000, 100_000)
    assert (runs[1:] & 3 != runs[:-1] & 3).all()     # maximal runs, as CompressCigar prints them
    lead = int(runs[0])
    assert int(b.target_begins()[0]) == (3 if (lead & 3) == 3 and (lead >> 2) <= 9 else 1)
    assert score == oracle.score_linear(q, t, 0, 1, -1, -1)


@pytest.mark.parametrize("length,target_len,n", [(20_000, 22_000, 6), (40_000, 44_000, 4)])
def test_long_pairs_on_the_packed_wavefront_kernels(oracle, length, target_len, n):
    """Global and Semiglobal pairs of 20 kbp x 22 kbp and 40 kbp x 44 kbp (matrix values from -84 000 along the
    borders to +30 000 on the path: far outside 15 bits) on the packed 16-bit wavefront kernel, which keeps every lane's scores relative
    to a base of its own, in its forms: one wavefront per two pairs, one per (two pairs, strip), and one per (two
    pairs, strip, segment of 1024 target rows) as a launch with more strips than the GPU holds wavefronts runs.  Every
    score equals the CPU checker's linear-memory scorer, every CIGAR re-scores to it, consumes the aligned spans
    and has M on equal and X on unequal symbols; at 20 kbp the first pair of each type equals the CPU checker's
    full-matrix result verbatim.  Pairs of a batch differ in length (the two pairs of a wavefront then end in
    different strips), and one of them is unrelated to its target."""
    import torch
    from sequencealigner_amd import align, capi, synth
    rng = np.random.default_rng(length)
    qs, ts = [], []
    for p in range(n):
        t = synth.random_bases(rng, target_len - 700 * p)
        lead = 900 if p % 2 else 0
        q = synth.fit_length(rng, synth.edit(rng, t[lead:lead + length], 0.04, 0.03, 0.03), length - 1100 * p)
        if p == 3:
            q = synth.random_bases(rng, q.shape[0])
        qs.append(bytes(q))
        ts.append(bytes(t))
    qb, qo = synth.to_blob(qs)
    tb, to = synth.to_blob(ts)
    for typ in (0, 2):
        want = oracle.score_linear_batch(qb, qo, tb, to, typ, 1, -1, -1, 8)
        full = oracle.align(qs[0], ts[0], typ, 1, -1, -1) if length <= 20_000 else None
        for name, coop, pk, seg in (("wave_pk16", 0, -1, -1), ("wave_pk16_coop", 1, 1, -1), ("wave_pk16_coop", 1, 1, 1024)):
            capi.set_option("force_coop", coop)
            capi.set_option("coop_pk", pk)
            capi.set_option("coop_seg_rows", seg)
            try:
                b = align.DeviceAlignBatch(qs, ts, typ, 1, -1, -1)
                assert b.kernel_name() == name
                b.run()
                torch.cuda.synchronize()
            finally:
                capi.set_option("force_coop", -1)
                capi.set_option("coop_pk", -1)
                capi.set_option("coop_seg_rows", -1)
            scores = b.scores()
            assert np.array_equal(scores, want), (typ, name, seg)
            nruns = b.d_nruns[:n].cpu().numpy()
            runs = b.d_runs.cpu().numpy().view(np.uint32).reshape(n, b.stride)
            rs, qu, tu = _rescore(runs, nruns, b.stride, 1, -1, -1)
            assert np.array_equal(rs, scores), (typ, name)
            for p in range(n):
                pr = runs[p, b.stride - nruns[p]:]
                assert (pr[1:] & 3 != pr[:-1] & 3).all()
                assert int(qu[p]) == len(qs[p])
                if typ == 0:
                    assert int(tu[p]) == len(ts[p]) and _walk(pr, qs[p], ts[p], 0, 0) == (len(qs[p]), len(ts[p]))
                else:
                    assert 0 < int(tu[p]) <= len(ts[p])
            if full is not None:
                assert (int(scores[0]), b.cigars([0])[0], int(b.target_begins()[0])) == full, (typ, name)
            del b, runs
