"""Times seed chaining on the GPU: 4.6 Mbp reference, reads of 10 kbp drawn from it with 10 % edits
(k = 15, w = 5, f = 0.001), minimizers and chains computed on the device; CPU restatement timed on a
sample for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from sequencealigner_amd import chain, minimize, synth
import cpu_checker

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
L = 10_000
ref = synth.genome(4_600_000, seed=4)
rng = np.random.default_rng(9)
starts = rng.integers(0, len(ref) - L, n_reads)
reads = np.stack([ref[s:s + L] for s in starts])
sub = rng.random(reads.shape) < 0.10
reads = np.where(sub, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, reads.shape)], reads).astype(np.uint8)

def lists(seqs):
    b = minimize.DeviceMinimizeBatch(seqs, 15, 5)
    b.run(); torch.cuda.synchronize()
    off = b.d_out_off.clone(); n = int(off[-1].item())
    return b.d_out[:3 * n].clone().reshape(n, 3), off

d_ref, _ = lists([ref])
d_reads, d_off = lists(reads)
print("reference minimizers", d_ref.shape[0], "read minimizers", d_reads.shape[0], flush=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out, nm = chain.chain_batch(d_ref, d_reads, d_off, 0.001, want_matches=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("chain %d reads: %.2f ms, %.2f M reads/s, %.1f M matches, mean chain %.1f, span ok %.3f" % (
        n_reads, dt * 1e3, n_reads / dt / 1e6, nm.sum() / 1e6, out[:, 0].mean(),
        (np.abs(out[:, 1] - starts) < 200).mean()), flush=True)
chk = cpu_checker.restatement()
sample = min(n_reads, 2000)
offs = d_off.cpu().numpy().view(np.uint64)
t0 = time.time()
want, _ = chk.chain_batch(d_ref.cpu().numpy().view(np.uint32), d_reads.cpu().numpy().view(np.uint32)[:int(offs[sample])],
                          offs[:sample + 1], 0.001)
dt = time.time() - t0
print("CPU restatement, 1 thread: %d reads in %.2f s (%.0f reads/s, index build included); equal: %s" % (
    sample, dt, sample / dt, np.array_equal(want, out[:sample])), flush=True)
