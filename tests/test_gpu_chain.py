"""GPU parity of the seed-chaining step (index, frequency filter, match, longest increasing subsequence)
against the CPU restatement of the mapper's functions (reference src/mapper.cpp:172-352), through the
C-ABI.  Those functions are static inside mapper.cpp, which cannot be built from the reference's sources
alone: the restatement is the only checker here (parity unpinned)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mutate(rng, s, rate):
    out = bytearray()
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))
        elif x < rate:
            out.append(c)
            out.append(rng.choice(b"ACGT"))
            continue
        out.append(c)
    return bytes(out)


def _lists(seqs, k, w):
    import torch
    from sequencealigner_amd import minimize
    b = minimize.DeviceMinimizeBatch(seqs, k, w)
    b.run()
    torch.cuda.synchronize()
    off = b.d_out_off.clone()
    n = int(off[-1].item())
    return b.d_out[:3 * n].clone().reshape(n, 3), off


@pytest.mark.parametrize("f", [0.0, 0.001, 0.2])
def test_chain_vs_restatement(oracle, f):
    import random
    from sequencealigner_amd import chain, synth
    rng = random.Random(23)
    ref = bytes(synth.genome(150_000, seed=51))
    ref = ref[:60_000] + ref[20_000:30_000] + b"ACGTTGCA" * 400 + ref[60_000:]  # a repeat and a tandem stretch
    reads = []
    for _ in range(120):
        L = rng.randint(300, 6000)
        p = rng.randint(0, len(ref) - L)
        reads.append(_mutate(rng, ref[p:p + L], rng.choice([0.0, 0.05, 0.15])))
    reads.append(bytes(synth.genome(2000, seed=52)))       # unrelated: few or no matches
    reads.append(ref[61_000:61_040])                       # a handful of minimizers
    reads.append(b"ACGTTGCA" * 60)                         # only the tandem stretch
    k, w = 15, 5
    d_ref, _ = _lists([ref], k, w)
    d_reads, d_off = _lists(reads, k, w)
    got, nm = chain.chain_batch(d_ref, d_reads, d_off, f, want_matches=True)
    want, wnm = oracle.chain_batch(d_ref.cpu().numpy().view(np.uint32), d_reads.cpu().numpy().view(np.uint32),
                                   d_off.cpu().numpy().view(np.uint64), f)
    assert np.array_equal(nm.astype(np.uint64), wnm)
    assert np.array_equal(got, want)
    assert (got[:100, 0] > 3).any()


def test_chain_small_k_many_matches(oracle):
    """Short k-mers: every value occurs many times, match lists are long and full of equal positions."""
    from sequencealigner_amd import chain, synth
    ref = bytes(synth.genome(5_000, seed=61))
    reads = [ref[100:700], ref[2000:2400], bytes(synth.genome(500, seed=62))]
    d_ref, _ = _lists([ref], 4, 3)
    d_reads, d_off = _lists(reads, 4, 3)
    for f in (0.0, 0.5):
        got, nm = chain.chain_batch(d_ref, d_reads, d_off, f, want_matches=True)
        want, wnm = oracle.chain_batch(d_ref.cpu().numpy().view(np.uint32), d_reads.cpu().numpy().view(np.uint32),
                                       d_off.cpu().numpy().view(np.uint64), f)
        assert np.array_equal(nm.astype(np.uint64), wnm) and np.array_equal(got, want)
