"""Synthetic FASTA / FASTQ file contents for the ingest tests (shared by the CPU and the GPU tests)."""
import numpy as np


def _bases(rng, n, alphabet=b"ACGT"):
    return bytes(np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), n)])


def fasta(rng, lengths, width=60, eol=b"\n", name_extra=b" some description > with a bracket", last_eol=True,
          alphabet=b"ACGT", header_len=0):
    """-> (file bytes, [names], [sequences]).  header_len pads every header line to at least that many
    bytes (headers longer than a tile of the device parser)."""
    out, names, seqs = bytearray(), [], []
    for i, n in enumerate(lengths):
        name = b"rec%d" % i
        s = _bases(rng, n, alphabet)
        names.append(name)
        seqs.append(s)
        head = b">" + name + name_extra
        if header_len > len(head):
            head += b" " + b"x>" * ((header_len - len(head)) // 2)
        out += head + eol
        w = width if width else max(n, 1)
        for j in range(0, n, w):
            out += s[j:j + w] + eol
    if not last_eol and out.endswith(eol):
        del out[-len(eol):]
    return bytes(out), names, seqs


def fastq(rng, lengths, eol=b"\n", last_eol=True, name_extra=b" len=1"):
    out, names, seqs = bytearray(), [], []
    for i, n in enumerate(lengths):
        name = b"read%d" % i
        s = _bases(rng, n)
        # qualities drawn from the whole printable range, '@' and '+' at line starts included
        q = bytes(rng.integers(33, 127, n, dtype=np.uint8))
        if i % 3 == 0:
            q = b"@" + q[1:]
        elif i % 3 == 1:
            q = b"+" + q[1:]
        names.append(name)
        seqs.append(s)
        out += b"@" + name + name_extra + eol + s + eol + b"+" + (name if i % 2 else b"") + eol + q + eol
    if not last_eol and out.endswith(eol):
        del out[-len(eol):]
    return bytes(out), names, seqs
