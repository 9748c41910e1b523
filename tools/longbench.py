"""Times the long-pair shapes: semi-global 10 kbp x 11 kbp (a sample of the 200k-pair config) and one
global 100 kbp x 100 kbp pair; checks every CIGAR re-scores to the reported score."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sequencealigner_amd import align, synth

def rescore(b, n, match=1, mismatch=-1, gap=-1):
    nruns = b.d_nruns[:n].cpu().numpy()
    scores = b.scores()
    ok = True
    for p in range(min(n, 16)):
        e = (p + 1) * b.stride
        runs = b.d_runs[e - int(nruns[p]):e].cpu().numpy().view(np.uint32)
        ln = (runs >> 2).astype(np.int64); op = runs & 3
        s = int((ln * np.where(op == 0, match, np.where(op == 1, mismatch, gap))).sum())
        qa = int(ln[(op == 0) | (op == 1) | (op == 2)].sum()); ta = int(ln[(op == 0) | (op == 1) | (op == 3)].sum())
        ok &= (s == int(scores[p]))
        if p < 2: print("   pair", p, "score", int(scores[p]), "rescored", s, "runs", len(runs), "q used", qa, "t used", ta)
    return ok

which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "sg"):
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    q, t = synth.pairs_semiglobal_long(n, seed=3)
    b = align.DeviceAlignBatch(q, t, 2, 1, -1, -1)
    print("semi-global", n, "pairs 10k x 11k kernel", b.kernel_name(), "workspace GB", b.workspace_bytes / 1e9, flush=True)
    import ctypes as C
    from sequencealigner_amd import capi
    lib = capi.lib()
    for it in range(2):
        lib.sa_align_profile(1)
        torch.cuda.synchronize(); t0 = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t0
        ms = (C.c_double * 3)(); cnt = (C.c_uint64 * 3)()
        lib.sa_align_profile_read(ms, cnt); lib.sa_align_profile(0)
        print("  run %.1f ms  %.1f GCUPS  (fill %.1f ms, traceback %.1f ms)" % (dt * 1e3, b.cells / dt / 1e9, ms[1], ms[2]), flush=True)
    print("  cigars re-score:", rescore(b, n))
    del b; torch.cuda.empty_cache()
if which in ("both", "nw"):
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    q, t = synth.pair_global_long(L, seed=5)
    b = align.DeviceAlignBatch([q], [t], 0, 1, -1, -1)
    print("global 1 pair", L, "x", L, "kernel", b.kernel_name(), "workspace GB", b.workspace_bytes / 1e9, flush=True)
    import ctypes as C
    from sequencealigner_amd import capi
    lib = capi.lib()
    for it in range(2):
        lib.sa_align_profile(1)
        torch.cuda.synchronize(); t0 = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t0
        ms = (C.c_double * 3)(); cnt = (C.c_uint64 * 3)()
        lib.sa_align_profile_read(ms, cnt); lib.sa_align_profile(0)
        print("  run %.1f ms  %.2f GCUPS  (fill %.1f ms, traceback %.1f ms)" % (dt * 1e3, b.cells / dt / 1e9, ms[1], ms[2]), flush=True)
    print("  cigar re-scores:", rescore(b, 1))
