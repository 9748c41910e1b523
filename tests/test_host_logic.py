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
