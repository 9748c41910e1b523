"""Compares batched MinimizeBinary results with the CPU checker read by read and prints what differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import cpu_checker
from sequencealigner_amd import capi, minimize, synth
chk = cpu_checker.restatement()
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if chunk:
    capi.set_option("min_chunk", chunk)
reads = synth.reads(300, 10_000, seed=74)
seqs = [bytes(r) for r in reads]
if len(sys.argv) > 2:   # one long sequence in the batch: the others take the several-kernel sort path
    seqs.append(bytes(synth.genome(60_000, seed=5)))
res = minimize.minimize_batch(seqs, 15, 5)
bad = 0
for i in range(300):
    want = chk.minimize_binary_array(bytes(reads[i]), 15, 5)
    got = res[i]
    if not np.array_equal(got, want):
        bad += 1
        ws = set(map(tuple, want.tolist())); gs = set(map(tuple, got.tolist()))
        if bad <= 8:
            print("read", i, "missing", sorted(ws - gs, key=lambda t: t[1])[:6], "extra", sorted(gs - ws, key=lambda t: t[1])[:6], flush=True)
print("chunk option", chunk, "reads that differ:", bad)
