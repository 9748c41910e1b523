"""Directed GPU tests at the range guards of the packed 16-bit kernels.  The dispatcher may use them only while
every value the DP forms fits their score window (15 bits for Global / Semiglobal in the lane kernel, a bound on
the difference of neighbouring cells in the wavefront kernel; Local's best-cell key leaves 9 bits in the lane
kernel and 11 in the wavefront kernel); one base or one score unit beyond, it must fall back to the 32-bit kernels.  For each guard: a batch whose
largest admissible value sits exactly at the window edge must run on the packed kernel and equal the CPU
checker, and the batch one step beyond must run on the 32-bit kernel and equal it too.  Contents that reach the
extremes: identical sequences (highest scores), sequences without a common symbol (lowest), related random ones."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(Q, T, seed):
    """Four pairs of query length Q, target length T: identical prefix, no common symbol, two related random."""
    from sequencealigner_amd import synth
    rng = np.random.default_rng(seed)
    t = synth.random_bases(rng, T)
    same = (np.resize(t, Q) if Q > T else t[:Q].copy(), t)
    apart = (np.full(Q, ord("A"), np.uint8), np.full(T, ord("C"), np.uint8))
    out = [same, apart]
    for _ in range(2):
        tt = synth.random_bases(rng, T)
        q = synth.fit_length(rng, synth.edit(rng, tt[:min(Q, T)], 0.05, 0.03, 0.03), Q)
        out.append((q, tt))
    return [bytes(q) for q, _ in out], [bytes(t) for _, t in out]


def _check(oracle, family, coop, typ, scheme, Q, T, want_kernel, seed):
    import torch
    from sequencealigner_amd import align, capi
    m, x, g = scheme
    qs, ts = _pairs(Q, T, seed)
    capi.set_option("force_family", family)
    capi.set_option("force_coop", coop)
    try:
        b = align.DeviceAlignBatch(qs, ts, typ, m, x, g)
        assert b.kernel_name() == want_kernel, (b.kernel_name(), typ, scheme, Q, T)
        b.run()
        torch.cuda.synchronize()
        got = list(zip(b.scores().tolist(), b.cigars(), b.target_begins().tolist()))
    finally:
        capi.set_option("force_family", 0)
        capi.set_option("force_coop", -1)
    for i, (q, t) in enumerate(zip(qs, ts)):
        want = oracle.align(q, t, typ, m, x, g)
        if typ != 0 and want[0] <= 0:
            continue   # outside the reference's defined behaviour
        assert got[i] == want, (typ, scheme, Q, T, i, got[i][0], want[0])


# (type, scheme, T, largest Q the packed kernel may take), from the window arithmetic noted at each case; the
# test fails if the dispatcher's predicate is off by one in either direction
LANE_CASES = [
    # Local 1/-1/-1: bias 2, scores up to min(Q, t_pad): Q + 2 + 2 + 1 <= 511
    (1, (1, -1, -1), 528, 506),
    # Global 1/-15/-15, T = t_pad = 1056: 15 (Q + 1056) + 16 + min(Q, 1056) + 16 + 1 <= 32767
    (0, (1, -15, -15), 1056, 1055),
    # Semiglobal 1/-15/-15, T = 2064: 15 Q + 16 + Q + 16 + 1 <= 32767
    (2, (1, -15, -15), 2064, 2045),
]


@pytest.mark.parametrize("typ,scheme,T,q_edge", LANE_CASES)
def test_packed_lane_kernel_window_edge(oracle, typ, scheme, T, q_edge):
    _check(oracle, 1, -1, typ, scheme, q_edge, T, "lane_pk16", seed=typ)
    _check(oracle, 1, -1, typ, scheme, q_edge + 1, T, "lane_i32", seed=10 + typ)


WAVE_CASES = [
    # Global / Semiglobal keep scores relative to per-lane bases: no bound on the length, but neighbouring cells
    # may differ by at most 64, max(-gap, match - gap) <= 64 ...
    (0, (1, -3, -63), 1500, 1700, ("scheme", (1, -3, -64))),
    (2, (7, -3, -57), 1500, 1700, ("scheme", (8, -3, -57))),
    # ... and absolute scores travel between strips as 24-bit numbers: 1000 (Q + T + 2048) < 2^23
    (2, (1, -1000, -3), 3000, 3340, ("T", 3341)),
    # Local keeps scores times 16 in 15 bits.  1/-1/-1: bias 2, min(Q, T) + 2 + 2 + 1 <= 2047
    (1, (1, -1, -1), 2042, 2100, ("Q", 2043)),
    # Local 2/-3/-2: bias 4, 2 min(Q, T) + 4 + 4 + 1 <= 2047
    (1, (2, -3, -2), 1500, 1019, ("T", 1020)),
]


@pytest.mark.parametrize("typ,scheme,Q,T,beyond", WAVE_CASES)
def test_packed_wavefront_kernel_window_edge(oracle, typ, scheme, Q, T, beyond):
    _check(oracle, 2, 0, typ, scheme, Q, T, "wave_pk16", seed=20 + typ)
    what, value = beyond
    if what == "scheme":
        _check(oracle, 2, 0, typ, value, Q, T, "wave_i32", seed=30 + typ)
    elif what == "T":
        _check(oracle, 2, 0, typ, scheme, Q, value, "wave_i32", seed=30 + typ)
    else:
        _check(oracle, 2, 0, typ, scheme, value, T, "wave_i32", seed=30 + typ)


def test_conditions_the_packed_kernels_never_take(oracle):
    """Scorings outside the packed kernels' preconditions (gap > 0, mismatch > match, match < gap; Local with
    gap = 0 or mismatch = 0) run on the 32-bit kernels whatever the sizes."""
    for typ, scheme in ((0, (1, -1, 1)), (2, (1, 2, -1)), (0, (-3, -4, -2)), (1, (1, -1, 0)), (1, (2, 0, -1))):
        for family, name in ((1, "lane_i32"), (2, "wave_i32")):
            _check(oracle, family, 0, typ, scheme, 90, 120, name, seed=40 + typ)
