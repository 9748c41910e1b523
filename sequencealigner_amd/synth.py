"""Deterministic synthetic inputs of the benchmark shapes (uniform ACGT targets, edited queries).

One generator feeds the GPU path, the CPU checker and the fixtures, so all of them see identical bytes.
All sequences are uint8 arrays of ASCII 'A','C','G','T'.
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_bases(rng, shape):
    return BASES[rng.integers(0, 4, size=shape, dtype=np.uint8)]


def substitute(rng, seq, rate):
    """Replace each base with a DIFFERENT base with probability `rate` (seq: uint8 array, any shape)."""
    idx = np.searchsorted(BASES_SORTED, seq)
    hit = rng.random(seq.shape) < rate
    shift = rng.integers(1, 4, size=seq.shape, dtype=np.int64)
    new_idx = np.where(hit, (idx + shift) % 4, idx)
    return BASES_SORTED[new_idx]


BASES_SORTED = np.sort(BASES)


def edit(rng, seq, sub, ins, dele):
    """ONT-like edits on one 1-D sequence: per base, delete it (prob dele), precede it with a random
    inserted base (prob ins), substitute it (prob sub)."""
    n = seq.shape[0]
    u = rng.random(n)
    deleted = u < dele
    inserted = (u >= dele) & (u < dele + ins)
    body = substitute(rng, seq, sub)
    counts = np.where(deleted, 0, 1) + inserted.astype(np.int64)
    out = np.repeat(body, counts)
    # first copy of an inserted base's pair is the inserted (random) one
    starts = np.cumsum(counts) - counts
    ins_pos = starts[inserted]
    out[ins_pos] = random_bases(rng, ins_pos.shape[0])
    return out


def fit_length(rng, seq, length):
    if seq.shape[0] >= length:
        return seq[:length]
    return np.concatenate([seq, random_bases(rng, length - seq.shape[0])])


def pairs_short_read_window(n, seed=2, read_len=150, window_len=300, sub=0.02):
    """Config 'Smith-Waterman local, reads x windows': target window = uniform ACGT, query = a
    read_len substring at a random offset with `sub` substitutions.
    -> (queries (n, read_len) uint8, targets (n, window_len) uint8)"""
    rng = np.random.default_rng(seed)
    targets = random_bases(rng, (n, window_len))
    off = rng.integers(0, window_len - read_len + 1, size=n)
    cols = off[:, None] + np.arange(read_len)[None, :]
    queries = np.take_along_axis(targets, cols, axis=1)
    queries = substitute(rng, queries, sub)
    return np.ascontiguousarray(queries), np.ascontiguousarray(targets)


def pairs_global_1k(n, seed=1, length=1000, sub=0.05, ins=0.025, dele=0.025):
    """Config 'Needleman-Wunsch global 1 kbp x 1 kbp': query = edited target, fixed to `length`."""
    rng = np.random.default_rng(seed)
    targets = random_bases(rng, (n, length))
    queries = np.empty((n, length), dtype=np.uint8)
    for p in range(n):
        queries[p] = fit_length(rng, edit(rng, targets[p], sub, ins, dele), length)
    return queries, targets


def pairs_semiglobal_long(n, seed=3, read_len=10000, window_len=11000, lead=500, sub=0.04, ins=0.03,
                          dele=0.03):
    """Config 'semi-global ONT-like': query = target[lead : lead+~read_len] with ~10 % edits, fixed
    to read_len."""
    rng = np.random.default_rng(seed)
    targets = random_bases(rng, (n, window_len))
    queries = np.empty((n, read_len), dtype=np.uint8)
    for p in range(n):
        q = edit(rng, targets[p, lead:lead + read_len], sub, ins, dele)
        queries[p] = fit_length(rng, q, read_len)
    return queries, targets


def pair_global_long(length, seed=5, err=0.10):
    """Config 'long-pair global': one pair, query = target with ~err ONT-like edits."""
    rng = np.random.default_rng(seed)
    target = random_bases(rng, length)
    query = fit_length(rng, edit(rng, target, err * 0.4, err * 0.3, err * 0.3), length)
    return query, target


def genome(length, seed=4):
    return random_bases(np.random.default_rng(seed), length)


def reads(n, length, seed=6):
    return random_bases(np.random.default_rng(seed), (n, length))


def to_blob(rows):
    """(n, L) uint8 array or list of 1-D uint8 arrays -> (blob uint8, offsets uint64[n+1])"""
    if isinstance(rows, np.ndarray) and rows.ndim == 2:
        n, L = rows.shape
        return np.ascontiguousarray(rows).reshape(-1), (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    lens = np.array([len(r) for r in rows], dtype=np.uint64)
    off = np.zeros(len(rows) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    rows = [np.frombuffer(bytes(r), dtype=np.uint8) if isinstance(r, (bytes, bytearray, str)) else
            np.asarray(r, dtype=np.uint8) for r in rows]
    blob = np.concatenate(rows) if len(rows) else np.zeros(0, np.uint8)
    return blob, off
