"""GPU parity of the FASTA / FASTQ ingest (file bytes -> sequence blob, offsets, name spans) against the CPU
restatement of the parsing rules, through the C-ABI.  The reference's parser is bioparser, a submodule it
does not vendor, and it ships no sequence files: the restatement and the files generated here are the only
checkers (parity unpinned)."""
import numpy as np
import pytest

import seqfiles

pytestmark = pytest.mark.gpu


def _check(oracle, data, fmt, names=None, seqs=None):
    from sequencealigner_amd import capi, ingest
    want = oracle.parse(data, fmt)
    if want is None:  # both sides must reject the file
        assert names is None
        with pytest.raises(capi.SaError) as e:
            ingest.ingest_bytes(data, fmt)
        assert e.value.code == capi.SA_ERR_FORMAT
        return None
    rec = ingest.ingest_bytes(data, fmt)
    seq, off, ns, nl, qs = want
    n = len(off) - 1
    assert rec.n == n and rec.n_bases == int(off[-1])
    assert np.array_equal(rec.d_seq_off.cpu().numpy().astype(np.uint64), off)
    assert np.array_equal(rec.d_seq[:rec.n_bases].cpu().numpy(), seq)
    assert np.array_equal(rec.d_name_start.cpu().numpy().astype(np.uint64), ns)
    assert np.array_equal(rec.d_name_len.cpu().numpy().astype(np.uint32), nl)
    if fmt == ingest.FASTQ:
        assert np.array_equal(rec.d_qual_start.cpu().numpy().astype(np.uint64), qs)
    lens = (off[1:] - off[:-1]).astype(np.int64)
    if n:
        assert rec.min_len == lens.min() and rec.max_len == lens.max()
    if names is not None:
        assert rec.names(data) == names
        blob = bytes(seq)
        assert [blob[int(off[i]):int(off[i + 1])] for i in range(n)] == seqs
    return rec


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
@pytest.mark.parametrize("width", [60, 7, 0])
def test_fasta_vs_restatement(oracle, width, eol):
    rng = np.random.default_rng(11 + width)
    lengths = [int(x) for x in rng.integers(1, 3000, 300)] + [40000, 1, 16384, 16383, 5]
    data, names, seqs = seqfiles.fasta(rng, lengths, width=width, eol=eol)
    _check(oracle, data, 0, names, seqs)
    data, names, seqs = seqfiles.fasta(rng, lengths[:50], width=width, eol=eol, last_eol=False)
    _check(oracle, data, 0, names, seqs)


def test_fasta_edges_vs_restatement(oracle):
    rng = np.random.default_rng(5)
    # one tiny record, no final newline
    _check(oracle, b">a\nA", 0, [b"a"], [b"A"])
    # headers longer than a tile (16 KiB): whole tiles lie inside a header line
    data, names, seqs = seqfiles.fasta(rng, [100, 3, 20000, 7], header_len=40000)
    _check(oracle, data, 0, names, seqs)
    # one long record: the reference genome case, and lower case / other letters pass through
    data, names, seqs = seqfiles.fasta(rng, [1_000_003], width=80, alphabet=b"ACGTacgtNn")
    _check(oracle, data, 0, names, seqs)
    # file sizes around the tile size: the last header / line / byte straddles a tile
    for total in (16383, 16384, 16385, 32768 + 1):
        data, names, seqs = seqfiles.fasta(rng, [total - 200, 150], width=70)
        assert len(data) > total
        _check(oracle, data[:total], 0)
        _check(oracle, data[:total - 37], 0)


def test_fastq_vs_restatement(oracle):
    rng = np.random.default_rng(7)
    for eol in (b"\n", b"\r\n"):
        lengths = [int(x) for x in rng.integers(1, 400, 2000)] + [30000, 1, 16384]
        data, names, seqs = seqfiles.fastq(rng, lengths, eol=eol)
        _check(oracle, data, 1, names, seqs)
        data, names, seqs = seqfiles.fastq(rng, lengths[:10], eol=eol, last_eol=False)
        _check(oracle, data, 1, names, seqs)
    # very short lines: many line ends inside one 64-byte group
    data, names, seqs = seqfiles.fastq(rng, [1, 2, 1, 3] * 500, name_extra=b"")
    _check(oracle, data, 1, names, seqs)


def test_malformed_files_are_rejected(oracle):
    from sequencealigner_amd import capi, ingest
    rng = np.random.default_rng(3)
    good, _, _ = seqfiles.fastq(rng, [50, 60, 70])
    lines = good.split(b"\n")
    bad_fastq = [
        b"\n".join(lines[:-2]),                                   # a record of three lines
        good.replace(b"@read1", b"#read1"),                       # header without '@'
        b"\n".join(lines[:2] + [b"-"] + lines[3:]),               # third line without '+'
        b"\n".join(lines[:3] + [lines[3][:-1]] + lines[4:]),      # one quality short
        b"\n".join(lines[:1] + [b""] + lines[2:]),                # record without bases
    ]
    bad_fasta = [b"ACGT\n>a\nAC\n", b">\nACGT\n", b">a\n>b\nAC\n", b">a\nAC\n>b\n"]
    for fmt, cases in ((1, bad_fastq), (0, bad_fasta)):
        for data in cases:
            assert oracle.parse(data, fmt) is None
            with pytest.raises(capi.SaError) as e:
                ingest.ingest_bytes(data, fmt)
            assert e.value.code == capi.SA_ERR_FORMAT
    assert ingest.ingest_bytes(b"", 0).n == 0


def test_two_bit_round_trip():
    import torch
    from sequencealigner_amd import ingest
    rng = np.random.default_rng(9)
    for n in (1, 31, 32, 33, 1000, 100_003):
        raw = np.frombuffer(b"ACGTacgtNnRY-", dtype=np.uint8)[rng.integers(0, 13, n)]
        d = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
        d[:n] = torch.from_numpy(raw.copy()).cuda()
        packed, other = ingest.pack2bit(d, n)
        code = {ord("C"): 0, ord("A"): 1, ord("T"): 2, ord("G"): 3}
        p = packed.cpu().numpy().view(np.uint64)
        o = other.cpu().numpy().view(np.uint32)
        for i in rng.integers(0, n, 200):
            i = int(i)
            up = raw[i] & 0xDF
            assert ((int(o[i // 32]) >> (i % 32)) & 1) == (0 if up in code else 1)
            if up in code:
                assert ((int(p[i // 32]) >> (2 * (i % 32))) & 3) == code[up]
        back = ingest.unpack2bit(packed, other, n).cpu().numpy()
        want = np.where(np.isin(raw & 0xDF, list(code)), raw & 0xDF, ord("N")).astype(np.uint8)
        assert np.array_equal(back, want)


def test_ingested_reads_feed_the_minimizer_call(oracle):
    """File -> device records -> minimizers without the bases touching the host in between."""
    import torch
    from sequencealigner_amd import ingest, minimize
    rng = np.random.default_rng(21)
    data, names, seqs = seqfiles.fastq(rng, [int(x) for x in rng.integers(100, 2000, 300)])
    rec = ingest.ingest_bytes(data, 1)
    b = minimize.DeviceMinimizeBatch.from_device(rec.d_seq, rec.d_seq_off, rec.n, rec.n_bases, rec.min_len,
                                                 rec.max_len, 15, 5)
    b.run()
    torch.cuda.synchronize()
    got = b.results()
    for i in (0, 17, 299):
        assert [tuple(int(v) for v in t) for t in got[i]] == oracle.minimize_binary(seqs[i], 15, 5)


def test_files_of_tens_of_megabytes_vs_restatement(oracle):
    """Tens of thousands of tiles: the scan over the tile summaries runs over many blocks."""
    rng = np.random.default_rng(33)
    data, names, seqs = seqfiles.fasta(rng, [int(x) for x in rng.integers(1, 15000, 3000)], width=80)
    assert len(data) > 20_000_000
    _check(oracle, data, 0, names, seqs)
    data, names, seqs = seqfiles.fastq(rng, [int(x) for x in rng.integers(50, 9000, 2500)])
    assert len(data) > 20_000_000
    _check(oracle, data, 1, names, seqs)
