"""PCIe-inclusive rate of the host-buffer entry point sa_align_batch_host on the benchmark shape (Local,
query 150 x target 300): the caller hands over host pointers, gets scores, target_begin and (optionally) CIGAR
strings back.  Prints one JSON line per case.

    python tools/hostbench.py [--pairs 500000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sequencealigner_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    n = args.pairs
    q, t = synth.pairs_short_read_window(n)
    q, t = np.ascontiguousarray(q), np.ascontiguousarray(t)
    lib = capi.lib()
    qp = (C.c_char_p * n)(*[C.c_char_p(q.ctypes.data + i * q.shape[1]) for i in range(n)])
    tp = (C.c_char_p * n)(*[C.c_char_p(t.ctypes.data + i * t.shape[1]) for i in range(n)])
    ql = (C.c_uint32 * n)(*([q.shape[1]] * n))
    tl = (C.c_uint32 * n)(*([t.shape[1]] * n))
    scores = (C.c_int32 * n)()
    tb = (C.c_uint32 * n)()
    cells = float(n) * q.shape[1] * t.shape[1]
    # one pair per call: what the C++ drop-in makes of every Align() (reference src/mapper.cpp:423)
    one_q, one_t = (C.c_char_p * 1)(qp[0]), (C.c_char_p * 1)(tp[0])
    one_ql, one_tl = (C.c_uint32 * 1)(q.shape[1]), (C.c_uint32 * 1)(t.shape[1])
    cig1 = (C.c_void_p * 1)()
    for _ in range(20):
        capi.check(lib.sa_align_batch_host(one_q, one_ql, one_t, one_tl, 1, 1, 1, -1, -1, 1, scores, tb, cig1))
        lib.sa_free(cig1[0])
    t0 = time.perf_counter()
    calls = 500
    for _ in range(calls):
        capi.check(lib.sa_align_batch_host(one_q, one_ql, one_t, one_tl, 1, 1, 1, -1, -1, 1, scores, tb, cig1))
        lib.sa_free(cig1[0])
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "one pair per call, CIGAR", "calls": calls, "us_per_call": round(dt / calls * 1e6, 1)}),
          flush=True)
    for want_cigar in (False, True):
        best = 1e9
        for _ in range(args.reps + 1):
            cig = (C.c_void_p * n)() if want_cigar else None
            t0 = time.perf_counter()
            capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, n, 1, 1, -1, -1, 1, scores, tb, cig))
            dt = time.perf_counter() - t0
            best = min(best, dt)
            if want_cigar:
                for i in range(n):
                    lib.sa_free(cig[i])
        print(json.dumps({"case": "sa_align_batch_host sw_150x300", "pairs": n, "cigar_strings": want_cigar,
                          "seconds": round(best, 4), "GCUPS": round(cells / best / 1e9, 1),
                          "pairs_per_s": round(n / best)}), flush=True)


def minimizers():
    """sa_minimize_batch_host: 20 000 reads of 10 kbp in one call, and one read per call."""
    lib = capi.lib()
    n, L = 20000, 10000
    reads = np.ascontiguousarray(synth.reads(n, L, seed=8))
    sp = (C.c_char_p * n)(*[C.c_char_p(reads.ctypes.data + i * L) for i in range(n)])
    sl = (C.c_uint32 * n)(*([L] * n))
    outs = (C.c_void_p * n)()
    counts = (C.c_uint64 * n)()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        capi.check(lib.sa_minimize_batch_host(sp, sl, n, 15, 5, capi.SA_MIN_BINARY, 1, outs, counts))
        best = min(best, time.perf_counter() - t0)
        for i in range(n):
            lib.sa_free(outs[i])
    print(json.dumps({"case": "sa_minimize_batch_host 20000 x 10 kbp", "seconds": round(best, 4),
                      "Gbases_per_s": round(n * L / best / 1e9, 2)}), flush=True)
    one_s, one_l = (C.c_char_p * 1)(sp[0]), (C.c_uint32 * 1)(L)
    for _ in range(10):
        capi.check(lib.sa_minimize_batch_host(one_s, one_l, 1, 15, 5, capi.SA_MIN_BINARY, 1, outs, counts))
        lib.sa_free(outs[0])
    calls = 300
    t0 = time.perf_counter()
    for _ in range(calls):
        capi.check(lib.sa_minimize_batch_host(one_s, one_l, 1, 15, 5, capi.SA_MIN_BINARY, 1, outs, counts))
        lib.sa_free(outs[0])
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "one 10 kbp read per call (MinimizeBinary)", "calls": calls,
                      "us_per_call": round(dt / calls * 1e6, 1)}), flush=True)


if __name__ == "__main__":
    if "--minimizers" in sys.argv:
        sys.argv.remove("--minimizers")
        minimizers()
        sys.exit(0)
    main()
