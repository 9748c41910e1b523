"""The PAF record the reference's mapper prints for every mapped fragment (generatePAFString,
reference src/mapper.cpp:354-369, called at :431, :438, :441), built from the batch outputs.

Column order and separators follow the reference exactly, including its strand field, which is the three
characters " + " (a space on either side), its fixed mapping quality "0", and the order of its 10th and
11th values: the call sites pass the match length where the parameter is named alignmentScore and the
alignment block length after it (src/mapper.cpp:431).
"""


def paf_line(query_name, query_len, query_start, query_end, target_name, target_len, target_start,
             target_end, matches, block_len, cigar=None):
    fields = [query_name, str(query_len), str(query_start), str(query_end), " + ", target_name,
              str(target_len), str(target_start), str(target_end), str(matches), str(block_len), "0"]
    line = "\t".join(fields)
    if cigar is not None:
        line += "\t" + cigar
    return line + "\n"


def paf_lines(names, query_lens, spans, target_name, target_len, block_len, cigars, matches=None):
    """One record per aligned fragment from batch outputs: spans = (q_begin, q_end, t_begin, t_end) per
    fragment as chosen by the chaining step, block_len = DeviceAlignBatch.cigar_stats()[0]; matches
    defaults to the reference's value (always 0: src/mapper.cpp:392 looks for a lower-case 'm')."""
    out = []
    for p, name in enumerate(names):
        qb, qe, tb, te = spans[p]
        m = 0 if matches is None else int(matches[p])
        out.append(paf_line(name, int(query_lens[p]), qb, qe, target_name, target_len, tb, te, m,
                            int(block_len[p]), cigars[p] if cigars is not None else None))
    return out
