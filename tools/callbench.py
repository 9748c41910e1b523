"""Latency of one Global alignment per call through the host entry point: what the reference's mapper does per
read (reference src/mapper.cpp:423) when it is linked against the drop-in without batching.

    python tools/callbench.py
"""
import sys, time, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sequencealigner_amd import capi, synth
lib = capi.lib()
for L in (500, 1500, 3000, 4999):
    q, t = synth.pairs_global_1k(1, seed=3, length=L)
    q, t = np.ascontiguousarray(q[0]), np.ascontiguousarray(t[0])
    qp, tp = (C.c_char_p * 1)(q.ctypes.data), (C.c_char_p * 1)(t.ctypes.data)
    ql, tl = (C.c_uint32 * 1)(L), (C.c_uint32 * 1)(L)
    sc, tb, cg = (C.c_int32 * 1)(), (C.c_uint32 * 1)(), (C.c_void_p * 1)()
    for _ in range(5):
        capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, 1, 0, 1, -1, -1, 1, sc, tb, cg)); lib.sa_free(cg[0])
    t0 = time.perf_counter()
    for _ in range(50):
        capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, 1, 0, 1, -1, -1, 1, sc, tb, cg)); lib.sa_free(cg[0])
    dt = (time.perf_counter() - t0) / 50
    b = capi.AlignBatch(); b.n = 1; b.max_qlen = b.max_tlen = b.min_qlen = b.min_tlen = L; b.type = 0; b.match = 1; b.mismatch = -1; b.gap = -1
    print("Global %d x %d one pair per call: %.3f ms  (%s, score %d)" % (L, L, dt * 1e3, lib.sa_align_kernel_name(C.byref(b)).decode(), sc[0]), flush=True)
