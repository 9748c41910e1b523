"""`mapper`-compatible command line over the GPU pipeline: the argument surface, the stderr reports and the
PAF output of the reference's main() (reference src/mapper.cpp:446-529 options, :41-65 help, :67-69 version,
:71-105 fragment statistics, :107-132 and :580-627 demo alignment of two random short fragments, :632-653
reference index figures, :662-679 one PAF record per fragment, long fragments first).

    python -m sequencealigner_amd.cli [-a 0|1|2] [-m N] [-n N] [-g N] [-k N] [-w N] [-t N] [-f X] [-c X] [-h] [-v] \
        reference.fasta reads.fasta|reads.fq

Differences that are deliberate:
  * -t (threads of the mapper's pool) is accepted and ignored: reads are not dealt to host threads but run as
    batches on the GPU (the device batch queue that replaces the pool);
  * the mapper prints `target_begin` of its demo alignment from a variable it never assigns (:622-627) - here
    the value the alignment returned (0 without -c, as no traceback is run);
  * the mapper picks the two demo fragments with rand() seeded by the clock; --seed fixes the choice.
`c` takes an argument on the short form, as in the mapper's option string "vhc:a:m:n:g:k:w:t:f:" (:478), and
none as --cigar (:455).
"""
import getopt
import random
import sys

import numpy as np

PROJECT_VER = "1.0"   # the mapper's PROJECT_VER (reference CMakeLists.txt:4, src/config.h.in:4-5)

LONG_OPTIONS = ["version", "help", "alignment_type=", "match=", "nomatch=", "gap=", "cigar", "kmer_len=",
                "window_len=", "thread_num=", "frequent=", "seed="]

HELP_ROWS = [   # reference src/mapper.cpp:49-63
    ("v", "version", "no", ""), ("c", "alignment for frag size", "no", ""), ("", "< 5000, cigar printed", "", ""),
    ("k", "kmer len", "yes", "15"), ("w", "window len", "yes", "5"), ("a", "alignment type: ", "yes", "0"),
    ("", "0 for global", "", ""), ("", "1 for local", "", ""), ("", "2 for semi-global", "", ""),
    ("m", "match score", "yes", "1"), ("n", "mismatch score", "yes", "-1"), ("g", "gap penalty", "yes", "0"),
    ("t", "number of threads", "yes", "5"), ("f", "top frequent minimizer not", "yes", "0.001"),
    ("", "taken into consideration", "", ""),
]


def help_text():
    """The mapper's help table: four columns of widths 10, 30, 10 and 20 (reference src/mapper.cpp:42-46),
    cells cut to the column width, between rules."""
    widths = (10, 30, 10, 20)
    rule = "+" + "+".join("-" * w for w in widths) + "+\n"

    def row(cells):
        return "|" + "|".join(f"{c[:w]:>{w}}" for c, w in zip(cells, widths)) + "|\n"

    return rule + row(("command", "explanation", "parameter needed?", "default values")) + rule + \
        "".join(row(r) for r in HELP_ROWS) + rule


def stats_text(st):
    """PrintStats (reference src/mapper.cpp:96-104)."""
    avg = st["average"]
    return ("---------------Statistics---------------\n"
            f"Number of sequences: {st['count']}\n"
            f"Total length of all fragments: {st['total']}\n"
            f"Largest fragment: {st['longest_name']}\n"
            f"  length: {st['longest']}\n"
            f"Smallest fragment: {st['shortest_name']}\n"
            f"   length: {st['shortest']}\n"
            f"Average length: {avg:.6g}\n"          # operator<< of a double: 6 significant digits
            f"N50 length: {st['n50']}\n")


def alignment_text(qname, qlen, tname, tlen, type_, match, mismatch, gap, target_begin, score):
    """PrintAlignmentResult (reference src/mapper.cpp:118-129; it is handed the fragments' NAMES as query and
    target)."""
    return ("---------------Alignment---------------\n"
            f"Query: {qname}\nQuery len: {qlen}\nTarget: {tname}\nTarget len: {tlen}\n"
            f"Alignment type: {type_}\nmatch: {match}\nmismatch: {mismatch}\ngap: {gap}\n\n"
            "---------------Results---------------\n"
            f"alignment score: {score}\ntaget begin: {target_begin}\n\n")


def index_figures(values, f):
    """The reference index figures (reference src/mapper.cpp:198-222, :646-651) from the reference's minimizer
    values: distinct values, singletons, and the occurrence count at position size - 1 - (int)(size * f) of
    the (occurrences, value) pairs in increasing order."""
    distinct, counts = np.unique(np.asarray(values, dtype=np.uint32), return_counts=True)
    order = np.lexsort((distinct, counts))
    size = len(distinct)
    index = int(np.float32(size) * np.float32(f)) if size else 0   # (int) occurrences.size() * f, f a float
    most = int(counts[order][size - 1 - index]) if size else 0
    return size, int((counts == 1).sum()), most


def index_text(size, singletons, most):
    bar = "-" * 78 + "\n"
    frac = np.float32(singletons) / np.float32(size) if size else float("nan")
    return (bar + "In reference genome: \n" + f"num minimizer:{size}\n" + f"num singleton: {singletons}\n" +
            f"Singleton Fraction of refence genome: {frac:.6g}\n" +
            f"number of occurrences of the most frequent minimizer: {most}\n" + bar)


def parse_args(argv):
    """-> (options dict, positional arguments); options as the mapper's main() leaves them (:467-471)."""
    opt = dict(alignment_type=0, match=1, mismatch=-1, gap=0, cigar=False, k=15, w=5, threads=5, f=0.001,
               version=False, help=False, seed=None)
    pairs, rest = getopt.gnu_getopt(argv, "vhc:a:m:n:g:k:w:t:f:", LONG_OPTIONS)

    def atoi(text):   # C's atoi: leading integer, 0 when there is none
        import re
        m = re.match(r"\s*[+-]?\d+", text)
        return int(m.group()) if m else 0

    for name, value in pairs:
        if name in ("-v", "--version"):
            opt["version"] = True
        elif name in ("-h", "--help"):
            opt["help"] = True
        elif name in ("-a", "--alignment_type"):
            opt["alignment_type"] = atoi(value)
        elif name in ("-m", "--match"):
            opt["match"] = atoi(value)
        elif name in ("-n", "--nomatch"):
            opt["mismatch"] = atoi(value)
        elif name in ("-g", "--gap"):
            opt["gap"] = atoi(value)
        elif name in ("-c", "--cigar"):
            opt["cigar"] = True
        elif name in ("-k", "--kmer_len"):
            opt["k"] = atoi(value)
        elif name in ("-w", "--window_len"):
            opt["w"] = atoi(value)
        elif name in ("-t", "--thread_num"):
            opt["threads"] = atoi(value)
        elif name in ("-f", "--frequent"):
            opt["f"] = float(value)
        elif name == "--seed":
            opt["seed"] = int(value)
    return opt, rest


def main(argv=None, out=sys.stdout, err=sys.stderr):
    try:
        opt, rest = parse_args(sys.argv[1:] if argv is None else argv)
    except getopt.GetoptError as e:     # getopt_long prints its own message and the mapper goes on (:521-523)
        err.write(f"mapper: {e}\n")
        return 0
    if opt["version"]:
        err.write(f"v{PROJECT_VER}\n")
    if opt["help"]:
        out.write(help_text())
    if not rest:
        return 0
    if len(rest) < 2:
        err.write("mapper: a reference file and a fragment file are needed\n")
        return 0
    from . import align, ingest, mapper
    details = {}
    cigar = opt["cigar"]
    records = mapper.map_files(rest[0], rest[1], details=details, k=opt["k"], w=opt["w"], f=opt["f"],
                               match=opt["match"], mismatch=opt["mismatch"], gap=opt["gap"], cigar_needed=cigar)
    reads, names = details["reads"], details["read_names"]
    lens = reads.lengths()
    err.write(stats_text(ingest.stats(lens, names)))
    # the demo alignment of two different random short fragments (:580-627)
    short, _ = ingest.split_fragments(lens)
    if len(short) >= 2 and opt["alignment_type"] in (0, 1, 2):
        rng = random.Random(opt["seed"])
        i1, i2 = rng.sample(list(short), 2)
        off = reads.d_seq_off.cpu().numpy()
        seqs = [bytes(reads.d_seq[int(off[i]):int(off[i + 1])].cpu().numpy()) for i in (i1, i2)]
        score, _, tb = align.Align(seqs[0], len(seqs[0]), seqs[1], len(seqs[1]), opt["alignment_type"], opt["match"],
                                   opt["mismatch"], opt["gap"], want_cigar=cigar, want_target_begin=cigar)
        err.write(alignment_text(names[i1], len(seqs[0]), names[i2], len(seqs[1]), opt["alignment_type"],
                                 opt["match"], opt["mismatch"], opt["gap"], tb if cigar else 0, score))
    # the reference index figures (:646-653)
    from . import minimize
    ref = details["ref"]
    r_off = ref.d_seq_off[:2].cpu().numpy()
    ref_seq = bytes(ref.d_seq[int(r_off[0]):int(r_off[1])].cpu().numpy())
    values = minimize.minimize_binary(ref_seq, opt["k"], opt["w"])[:, 0]
    err.write(index_text(*index_figures(values, opt["f"])))
    for line in records:
        out.write(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
