#!/usr/bin/env python3
"""Benchmark of the batched alignment hot path on MI355X.

A step = one pass of batched Align() (DP fill + traceback + CIGAR run-length encoding) over one resident
batch of synthetic pairs.  Default workload: Smith-Waterman local, 1,000,000 pairs of 150 bp reads x
300 bp target windows, match 1 / mismatch -1 / gap -1, CIGAR and target_begin produced for every pair,
inputs resident in HBM before the timed region.  With N GPUs every rank aligns its own batch of the same
size (weak scaling, no data-path collective: pairs are independent); value = total cell updates of all
ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# nominal integer VALU peaks of one MI355X: 256 CUs x 64 lanes/clk x 2.4 GHz for 32-bit max/compare/select
# class operations, twice that for packed 16-bit (measured with tools/valu_peak.hip: 37.8 / 75.6 T/s)
PEAK_I32_TOPS = 256 * 64 * 2.4e9 / 1e12
PEAK_PK16_TOPS = 2 * PEAK_I32_TOPS
HBM_PEAK_GBS = 8000.0
OPS_PER_CELL = {0: 10, 1: 13, 2: 10}   # algorithmic integer lane-ops per cell update (Global/Local/Semiglobal)

WORKLOADS = {
    # name: (type, generator name, default pairs, q_len, t_len)
    "sw_150x300": (1, "pairs_short_read_window", 1_000_000, 150, 300),
    "nw_1kx1k": (0, "pairs_global_1k", 10_000, 1000, 1000),
}


def make_pairs(workload, n, seed):
    from sequencealigner_amd import synth
    typ, gen, _, ql, tl = WORKLOADS[workload]
    if gen == "pairs_short_read_window":
        return synth.pairs_short_read_window(n, seed=seed, read_len=ql, window_len=tl)
    return synth.pairs_global_1k(n, seed=seed, length=ql)


def cpu_baseline(q, t, typ, sample, threads):
    """Times the CPU path (the reference's own Align() where oracle/_ref is present, else the
    restatement) over `sample` pairs with a task farm of `threads` std::threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_checker
    from sequencealigner_amd import synth
    chk = cpu_checker.reference() or cpu_checker.restatement()
    qb, qo = synth.to_blob(q[:sample])
    tb, to = synth.to_blob(t[:sample])
    chk.align_batch(qb[: int(qo[64])], qo[:65], tb[: int(to[64])], to[:65], typ, 1, -1, -1, True, threads)
    t0 = time.perf_counter()
    scores, _ = chk.align_batch(qb, qo, tb, to, typ, 1, -1, -1, True, threads)
    dt = time.perf_counter() - t0
    cells = float(q.shape[1]) * t.shape[1] * sample
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": threads,
            "kind": "reference" if chk.kind == "reference" else "port",
            "sample": f"{sample} pairs of the same workload, CIGAR on, {dt:.1f} s"}, scores


def minimizer_leg(dev, steps, cpu_threads, with_cpu):
    """Secondary metric of the same hot path: MinimizeBinary k=15 w=5 on a 4.6 Mbp genome (one sequence)
    and on 1 Gbp of 10 kbp reads (100 000 sequences), inputs resident in HBM.  Returns a dict."""
    import torch
    from sequencealigner_amd import minimize, synth
    out = {"k": 15, "w": 5}
    for name, seqs in (("genome_4.6Mbp", [synth.genome(4_600_000, seed=4)]),
                       ("reads_100k_x_10kbp", synth.reads(100_000, 10_000, seed=6))):
        b = minimize.DeviceMinimizeBatch(seqs, 15, 5, device=dev)
        b.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        n_out = int(b.d_out_off[-1].item())
        bytes_alg = b.total_len + 12 * n_out     # 1 B/base read + 12 B per unique minimizer written
        # the per-strand state machine is integer compare/select work: per window 3 ops to roll the k-mer,
        # per strand 4 for the step and 2 per re-scanned k-mer of an expiring minimum, 6 to pick and emit
        ops = 3 + 2 * (4 + 2 * 5) + 6
        tops = b.windows / dt * ops / 1e12
        out[name] = {"ms": dt * 1e3, "windows_per_s": b.windows / dt, "minimizers_per_s": n_out / dt,
                     "unique_minimizers": n_out,
                     "roofline": {"bound": "valu_int", "achieved": tops, "peak": PEAK_I32_TOPS,
                                  "unit": "Tops/s (integer lane-ops; %d per window)" % ops,
                                  "frac": tops / PEAK_I32_TOPS,
                                  "hbm": {"achieved": bytes_alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": bytes_alg / dt / 1e9 / HBM_PEAK_GBS}}}
        if with_cpu and name.startswith("reads"):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import cpu_checker
            chk = cpu_checker.reference() or cpu_checker.restatement()
            sample = min(len(seqs), 100_000)
            blob = np.concatenate([np.ascontiguousarray(seqs[:sample]).reshape(sample, -1),
                                   np.zeros((sample, 1), np.uint8)], axis=1)   # NUL-terminated
            off = np.arange(sample, dtype=np.uint64) * np.uint64(blob.shape[1])
            lens = np.full(sample, seqs.shape[1], dtype=np.uint32)
            t0 = time.perf_counter()
            counts = chk.minimize_batch(blob.reshape(-1), off, lens, 15, 5, cpu_threads)
            cdt = time.perf_counter() - t0
            win = sample * (seqs.shape[1] - 18)
            out[name]["cpu_baseline"] = {
                "value": win / cdt, "unit": "windows/s", "cores": cpu_threads,
                "kind": "reference" if chk.kind == "reference" else "port",
                "sample": f"{sample} reads of the same workload, {cdt:.1f} s",
                "counts_equal_gpu": bool(np.array_equal(counts.astype(np.int64), b.counts()[:sample]))}
        del b
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sw_150x300", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the workload's size)")
    ap.add_argument("--no-cigar", action="store_true", help="score only (not the headline number)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="pairs for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-minimizers", action="store_true", help="skip the secondary minimizer metric")
    args = ap.parse_args()

    import torch
    from sequencealigner_amd import align, capi, dist as sdist

    world = sdist.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    dev = world.local_rank if world.is_dist else 0
    torch.cuda.set_device(dev)
    lib = capi.lib()

    typ, _, default_pairs, ql, tl = WORKLOADS[args.workload]
    n = args.pairs or default_pairs
    q, t = make_pairs(args.workload, n, seed=1000 + world.rank)
    batch = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, want_cigar=not args.no_cigar,
                                   want_target_begin=True, device=dev)
    stream = torch.cuda.current_stream()

    for _ in range(args.warmup):
        batch.run(stream)
    torch.cuda.synchronize()
    sdist.barrier(world)
    lib.sa_align_profile(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sdist.barrier(world)
    ms = (C.c_double * 3)()
    cnt = (C.c_uint64 * 3)()
    capi.check(lib.sa_align_profile_read(ms, cnt))
    lib.sa_align_profile(0)

    elapsed_max = sdist.reduce_max(world, elapsed)
    cells_rank = float(batch.cells)
    cells_all = sdist.reduce_sum(world, cells_rank)
    value = cells_all * args.steps / elapsed_max / 1e9

    # roofline of the dominant kernel (the DP fill), per launch, from HIP events on its own stream
    fill_launches = max(int(cnt[1]), 1)
    fill_ms = ms[1] / fill_launches
    cells_per_launch = cells_rank * args.steps / fill_launches
    fill_cups = cells_per_launch / (fill_ms * 1e-3)
    achieved_tops = fill_cups * OPS_PER_CELL[typ] / 1e12
    peak_tops = PEAK_PK16_TOPS if "pk16" in batch.kernel_name() else PEAK_I32_TOPS
    roofline = {
        "kernel": batch.kernel_name(), "bound": "valu_int", "achieved": achieved_tops,
        "peak": peak_tops, "unit": "Tops/s (integer lane-ops; %d per cell update)" % OPS_PER_CELL[typ],
        "frac": achieved_tops / peak_tops, "traffic": None,
        "fill_ms_per_launch": fill_ms, "fill_gcups": fill_cups / 1e9,
        "launches_per_step": {"pack": int(cnt[0]) // args.steps, "fill": int(cnt[1]) // args.steps,
                              "traceback": int(cnt[2]) // args.steps},
        "ms_per_step_by_kernel": {"pack": ms[0] / args.steps, "fill": ms[1] / args.steps,
                                  "traceback": ms[2] / args.steps},
        # direction bits written by the fill: 2 bits per cell
        "hbm": {"bound": "hbm", "achieved": cells_per_launch / 4 / (fill_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": cells_per_launch / 4 / (fill_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }

    # HBM traffic of the fill kernel per launch: from the committed counter passes of this same workload
    # (tools/profile_round.sh; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, in KiB, FETCH_SIZE doubled for
    # gfx950's half-counted coalesced reads).  Counters cannot be collected inside this process.
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_per_launch.json")
    if args.workload == "sw_150x300" and n == default_pairs and not args.no_cigar and os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file))
            k = next(v for name, v in pmc.items() if "fill_lane_pk16<1" in name)
            roofline["traffic"] = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
            roofline["traffic_unit"] = "bytes per fill launch (profiles/r01_pmc_per_launch.json)"
            roofline["algorithmic_bytes"] = cells_per_launch / 4
        except (StopIteration, KeyError, ValueError):
            pass

    out = None
    if world.rank == 0:
        cpu = None
        sample = args.cpu_sample
        if sample < 0:
            sample = min(n, 1_000_000 if args.workload == "sw_150x300" else 300)
        if sample > 0 and not world.is_dist:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            cpu, cpu_scores = cpu_baseline(q, t, typ, sample, threads)
            gpu_scores = batch.scores()[:sample]
            cpu["scores_equal_gpu"] = bool(np.array_equal(cpu_scores, gpu_scores))
        out = {
            "metric": "GCUPS (DP cell updates/s), batched Align() incl. traceback + CIGAR",
            "value": value, "unit": "GCUPS", "n_gpus": world.size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16x2",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} pairs/GPU, query {ql} x target {tl}, "
                                   f"{'Local' if typ == 1 else 'Global'}, match 1 mismatch -1 gap -1, "
                                   f"{'score only' if args.no_cigar else 'full CIGAR + target_begin'}",
                       "pairs_per_gpu": n, "cells_per_step_per_gpu": cells_rank},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        out["dtype"] = "u16x2" if "pk16" in batch.kernel_name() else "int32"
        if not args.no_minimizers and not world.is_dist:
            del batch
            torch.cuda.empty_cache()
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            out["minimizers"] = minimizer_leg(dev, 3, threads, sample != 0)
        print(json.dumps(out), flush=True)
    sdist.finalize(world)
    return out


if __name__ == "__main__":
    main()
