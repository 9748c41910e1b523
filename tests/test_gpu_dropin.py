"""GPU tests of the host-buffer entry points and the C++ drop-in: the reference's unit tests replayed
through the reference's own C++ prototypes (tests/cpp/test_dropin.cpp) and the multi-GPU batch queue
entry points called through the C-ABI with host pointers."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sequencealigner_amd", "csrc")


def test_cpp_dropin_replays_reference_unit_tests():
    exe = os.path.join(CSRC, "test_dropin")
    assert os.path.exists(exe), "build it with make -C sequencealigner_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all drop-in checks passed" in r.stdout


def _c_strings(seqs):
    arr = (C.c_char_p * len(seqs))(*seqs)
    lens = (C.c_uint32 * len(seqs))(*[len(s) for s in seqs])
    return arr, lens


def test_align_batch_host_ragged(oracle):
    from sequencealigner_amd import capi, synth
    lib = capi.lib()
    rng = np.random.default_rng(12)
    qs, ts = [], []
    for i in range(900):
        T = int(rng.integers(20, 400))
        t = bytes(synth.random_bases(rng, T))
        a = int(rng.integers(0, T // 2))
        q = bytes(synth.substitute(rng, np.frombuffer(t[a:a + int(rng.integers(10, T - a + 1))], np.uint8), 0.08))
        qs.append(q)
        ts.append(t)
    n = len(qs)
    for typ, (m, x, g) in ((1, (1, -1, -1)), (0, (2, -3, -2)), (2, (1, -1, -1))):
        qa, ql = _c_strings(qs)
        ta, tl = _c_strings(ts)
        scores = (C.c_int32 * n)()
        tb = (C.c_uint32 * n)()
        cig = (C.c_void_p * n)()
        capi.check(lib.sa_align_batch_host(qa, ql, ta, tl, n, typ, m, x, g, 0, scores, tb, cig))
        for i in range(n):
            want = oracle.align(qs[i], ts[i], typ, m, x, g)
            got_c = C.cast(cig[i], C.c_char_p).value.decode()
            lib.sa_free(cig[i])
            if typ == 1 and want[0] <= 0:
                continue
            if typ == 2 and want[0] <= 0:
                continue
            assert (scores[i], got_c, tb[i]) == want, (i, typ)


def test_minimize_batch_host_and_remove_duplicate(oracle):
    from sequencealigner_amd import capi, synth
    lib = capi.lib()
    seqs = [bytes(synth.genome(int(L), seed=100 + i)) for i, L in enumerate([19, 500, 12000, 90000, 33, 7000])]
    n = len(seqs)
    sa, sl = _c_strings(seqs)
    outs = (C.c_void_p * n)()
    counts = (C.c_uint64 * n)()
    capi.check(lib.sa_minimize_batch_host(sa, sl, n, 15, 5, capi.SA_MIN_BINARY, 0, outs, counts))
    for i in range(n):
        got = np.ctypeslib.as_array(C.cast(outs[i], C.POINTER(C.c_uint32)), shape=(counts[i], 3)).copy()
        lib.sa_free(outs[i])
        assert np.array_equal(got, oracle.minimize_binary_array(seqs[i], 15, 5)), i
    rng = np.random.default_rng(3)
    t = np.stack([rng.integers(0, 50, 5000), rng.integers(0, 30, 5000), rng.integers(0, 2, 5000)],
                 axis=1).astype(np.uint32)
    want_sorted, want_uniq = oracle.remove_duplicate(t.copy())
    buf = np.ascontiguousarray(t)
    out = np.zeros_like(buf)
    m = C.c_size_t(0)
    capi.check(lib.sa_remove_duplicate_host(buf.ctypes.data, len(buf), out.ctypes.data, C.byref(m)))
    assert [tuple(int(v) for v in r) for r in buf] == want_sorted
    assert [tuple(int(v) for v in r) for r in out[:m.value]] == want_uniq


def test_align_batch_host_large_batch_in_parts():
    """Above 65 536 pairs the host entry point cuts a GPU's share into parts that run side by side; scores,
    target_begin and CIGAR strings must equal those of the device-resident batch call."""
    from sequencealigner_amd import align, capi, synth
    lib = capi.lib()
    n = 70_000
    q, t = synth.pairs_short_read_window(n, seed=9, read_len=40, window_len=70)
    q, t = np.ascontiguousarray(q), np.ascontiguousarray(t)
    want_s, want_c, want_tb = align.align_batch(q, t, 1, 1, -1, -1)
    qp = (C.c_char_p * n)(*[C.c_char_p(q.ctypes.data + i * q.shape[1]) for i in range(n)])
    tp = (C.c_char_p * n)(*[C.c_char_p(t.ctypes.data + i * t.shape[1]) for i in range(n)])
    ql = (C.c_uint32 * n)(*([q.shape[1]] * n))
    tl = (C.c_uint32 * n)(*([t.shape[1]] * n))
    scores = (C.c_int32 * n)()
    tb = (C.c_uint32 * n)()
    cig = (C.c_void_p * n)()
    capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, n, 1, 1, -1, -1, 1, scores, tb, cig))
    assert np.array_equal(np.ctypeslib.as_array(scores), want_s)
    assert np.array_equal(np.ctypeslib.as_array(tb), want_tb)
    for i in range(n):
        if i % 7 == 0:
            assert C.cast(cig[i], C.c_char_p).value.decode() == want_c[i], i
        lib.sa_free(cig[i])


def test_align_batch_host_with_outliers(oracle):
    """Many short pairs and a few long, thin ones in one call: the entry point runs them as batches by size
    class; results come back in input order."""
    from sequencealigner_amd import capi, synth
    lib = capi.lib()
    rng = np.random.default_rng(21)
    qs, ts = [], []
    for i in range(400):
        long_one = i in (7, 250, 399)
        T = int(rng.integers(3000, 6000)) if long_one else int(rng.integers(100, 160))
        t = bytes(synth.random_bases(rng, T))
        a = int(rng.integers(0, T // 2))
        q = bytes(synth.substitute(rng, np.frombuffer(t[a:a + int(rng.integers(60, 100))], np.uint8), 0.05))
        qs.append(q)
        ts.append(t)
    n = len(qs)
    for typ in (0, 2):
        qa, ql = _c_strings(qs)
        ta, tl = _c_strings(ts)
        scores = (C.c_int32 * n)()
        tb = (C.c_uint32 * n)()
        cig = (C.c_void_p * n)()
        capi.check(lib.sa_align_batch_host(qa, ql, ta, tl, n, typ, 1, -1, -1, 0, scores, tb, cig))
        for i in range(n):
            got_c = C.cast(cig[i], C.c_char_p).value.decode()
            lib.sa_free(cig[i])
            want = oracle.align(qs[i], ts[i], typ, 1, -1, -1)
            if typ == 2 and want[0] <= 0:
                continue
            assert (scores[i], got_c, tb[i]) == want, (i, typ)


@pytest.fixture
def four_shards():
    """The host batch calls deal their units to 4 shards that all run on GPU 0 at the same time, each on its
    own host thread, stream and buffers: the multi-GPU path of the device batch queue on a one-GPU box."""
    from sequencealigner_amd import capi
    capi.set_option("fake_devices", 4)
    yield 4
    capi.set_option("fake_devices", 0)


def test_align_batch_host_over_several_shards(oracle, four_shards):
    from sequencealigner_amd import capi, synth
    lib = capi.lib()
    assert lib.sa_device_count() >= 1
    rng = np.random.default_rng(33)
    qs, ts = [], []
    for i in range(6000):
        T = int(rng.integers(30, 700)) if i % 50 else int(rng.integers(1500, 2600))
        t = bytes(synth.random_bases(rng, T))
        a = int(rng.integers(0, T // 2))
        q = bytes(synth.substitute(rng, np.frombuffer(t[a:a + int(rng.integers(10, T - a + 1))], np.uint8), 0.06))
        qs.append(q)
        ts.append(t)
    n = len(qs)
    qb, qo = synth.to_blob(qs)
    tbb, to = synth.to_blob(ts)
    for typ, (m, x, g) in ((0, (1, -1, -1)), (1, (2, -3, -2)), (2, (1, -1, -1))):
        qa, ql = _c_strings(qs)
        ta, tl = _c_strings(ts)
        scores = (C.c_int32 * n)()
        tb = (C.c_uint32 * n)()
        cig = (C.c_void_p * n)()
        capi.check(lib.sa_align_batch_host(qa, ql, ta, tl, n, typ, m, x, g, 0, scores, tb, cig))
        want_s, want_tb = oracle.align_batch(qb, qo, tbb, to, typ, m, x, g, True, 16)
        ok = np.ones(n, bool) if typ == 0 else want_s > 0   # the reference's defined envelope
        assert np.array_equal(np.ctypeslib.as_array(scores)[ok], want_s[ok]), typ
        assert np.array_equal(np.ctypeslib.as_array(tb)[ok], want_tb[ok]), typ
        for i in range(n):
            if i % 23 == 0 and ok[i]:
                assert C.cast(cig[i], C.c_char_p).value.decode() == oracle.align(qs[i], ts[i], typ, m, x, g)[1], i
            lib.sa_free(cig[i])
        # without CIGAR strings: Local still reports the start row, Global / Semiglobal report 1
        tb2 = (C.c_uint32 * n)()
        capi.check(lib.sa_align_batch_host(qa, ql, ta, tl, n, typ, m, x, g, 3, scores, tb2, None))
        assert np.array_equal(np.ctypeslib.as_array(scores)[ok], want_s[ok]), typ
        if typ == 1:
            assert np.array_equal(np.ctypeslib.as_array(tb2)[ok], want_tb[ok])
        else:
            assert (np.ctypeslib.as_array(tb2) == 1).all()


def test_minimize_batch_host_over_several_shards(oracle, four_shards):
    from sequencealigner_amd import capi, synth
    lib = capi.lib()
    rng = np.random.default_rng(5)
    seqs = [bytes(synth.genome(int(L), seed=300 + i))
            for i, L in enumerate(rng.integers(19, 20_000, size=700).tolist() + [150_000, 19])]
    n = len(seqs)
    sa, sl = _c_strings(seqs)
    outs = (C.c_void_p * n)()
    counts = (C.c_uint64 * n)()
    capi.check(lib.sa_minimize_batch_host(sa, sl, n, 15, 5, capi.SA_MIN_BINARY, 0, outs, counts))
    for i in range(n):
        got = np.ctypeslib.as_array(C.cast(outs[i], C.POINTER(C.c_uint32)), shape=(counts[i], 3)).copy()
        lib.sa_free(outs[i])
        assert np.array_equal(got, oracle.minimize_binary_array(seqs[i], 15, 5)), i
