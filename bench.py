#!/usr/bin/env python3
"""Benchmark of the batched alignment hot path on MI355X.

A step = one pass of batched Align() (DP fill + traceback + CIGAR run-length encoding) over one resident
batch of synthetic pairs.  Headline workload: Smith-Waterman local, 1,000,000 pairs of 150 bp reads x
300 bp target windows, match 1 / mismatch -1 / gap -1, CIGAR and target_begin produced for every pair,
inputs resident in HBM before the timed region.  With N GPUs every rank aligns its own batch of the same
size (weak scaling, no data-path collective: pairs are independent); value = total cell updates of all
ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Besides the headline it carries `legs`: the other benchmark shapes, each
timed the same way with its own roofline object - Needleman-Wunsch 10 000 pairs of 1 kbp x 1 kbp,
semi-global 10 kbp x 11 kbp (25 000 pairs per GPU, run in as many passes as the workspace needs; under
--gpus N a fixed set of 200 000 pairs is split over the ranks instead: strong scaling), one global
100 kbp x 100 kbp pair, `host_queue`: one call of the host-buffer entry point over every GPU the process sees -
and `minimizers` (MinimizeBinary k=15 w=5 on a 4.6 Mbp genome and on 1 Gbp of reads).
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
TYPE_NAME = {0: "Global", 1: "Local", 2: "Semiglobal"}

WORKLOADS = {
    # name: (type, default pairs, q_len, t_len)
    "sw_150x300": (1, 1_000_000, 150, 300),
    "nw_1kx1k": (0, 10_000, 1000, 1000),
}
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_per_launch.json")


def make_pairs(workload, n, seed):
    from sequencealigner_amd import synth
    _, _, ql, tl = WORKLOADS[workload]
    if workload == "sw_150x300":
        return synth.pairs_short_read_window(n, seed=seed, read_len=ql, window_len=tl)
    return synth.pairs_global_1k(n, seed=seed, length=ql)


def host_cores():
    """Threads the CPU baseline may use: every core this process is allowed to run on.  os.cpu_count() counts
    the whole machine; the share of a container shows in the affinity mask, the cgroup CPU quota and
    OMP_NUM_THREADS (which the GPU boxes set to the share of one GPU's slot)."""
    box = os.cpu_count() or 1
    limits = {"machine": box}
    try:
        limits["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            limits["cgroup_quota"] = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit() and int(os.environ["OMP_NUM_THREADS"]) > 0:
        limits["OMP_NUM_THREADS"] = int(os.environ["OMP_NUM_THREADS"])
    used = min(limits.values())
    return used, box, limits


def cpu_checker_lib():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_checker
    return cpu_checker, (cpu_checker.reference() or cpu_checker.restatement())


def cpu_baseline(q, t, typ, sample, threads, note=""):
    """Times the CPU path (the reference's own Align() where oracle/_ref is present, else the restatement)
    over the first `sample` pairs with a task farm of `threads` std::threads - the stand-in for the
    reference's thread pool (src/mapper.cpp:662-679)."""
    from sequencealigner_amd import synth
    _, chk = cpu_checker_lib()
    qb, qo = synth.to_blob(q[:sample])
    tb, to = synth.to_blob(t[:sample])
    warm = min(sample, 64 if q.shape[1] * t.shape[1] < 10 ** 6 else 1)
    chk.align_batch(qb[: int(qo[warm])], qo[:warm + 1], tb[: int(to[warm])], to[:warm + 1], typ, 1, -1, -1, True,
                    threads)
    t0 = time.perf_counter()
    scores, _ = chk.align_batch(qb, qo, tb, to, typ, 1, -1, -1, True, threads)
    dt = time.perf_counter() - t0
    cells = float(q.shape[1]) * t.shape[1] * sample
    used, box, limits = host_cores()
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": threads, "cores_used": threads,
            "cores_on_box": box, "core_limits": limits,
            "kind": "reference" if chk.kind == "reference" else "port",
            "sample": f"{sample} pairs of the same workload{note}, CIGAR on, {dt:.1f} s"}, scores


def pmc_entries(section, kernel_substr):
    """The entries of the committed counter summary (tools/summarize_pmc.py) for a kernel: one per (kernel, number
    of workgroups) met in the section's runs - "bench", "minimizers_reads", "minimizers_genome" are runs of
    their own, so a launch on the genome is never averaged with one on the reads."""
    try:
        pmc = json.load(open(PMC_FILE))
        return [v for name, v in pmc.get(section, {}).items() if kernel_substr in name]
    except (OSError, ValueError):
        return []


def pmc_commit():
    try:
        return json.load(open(PMC_FILE)).get("_meta", {}).get("commit")
    except (OSError, ValueError):
        return None


def pmc_traffic(section, kernel_substr, blocks=None):
    """HBM bytes per launch of a kernel from the committed counter passes (tools/profile_round.sh: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled for gfx950's half-counted wide reads).  blocks: the
    launch's number of workgroups (the entry with that grid; None: the mean over the kernel's launches).
    Counters cannot be collected inside this process; the file names the commit it was collected at."""
    es = pmc_entries(section, kernel_substr)
    if blocks is not None:
        es = [e for e in es if e.get("blocks") == blocks]
    es = [e for e in es if "FETCH_SIZE" in e and "WRITE_SIZE" in e]
    if not es:
        return None
    n = sum(e["dispatches"] for e in es)
    return sum((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0 * e["dispatches"] for e in es) / n


def timed_alignment(lib, batch, steps, warmup, world=None, sdist=None):
    """warmup + steps runs of one resident batch; wall time per step (max over ranks when distributed) and
    HIP-event time per kernel class on the batch's own stream."""
    import torch
    from sequencealigner_amd import capi
    stream = torch.cuda.current_stream()
    for _ in range(warmup):
        batch.run(stream)
    torch.cuda.synchronize()
    if world is not None:
        sdist.barrier(world)
    lib.sa_align_profile(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world is not None:
        sdist.barrier(world)
    ms = (C.c_double * 3)()
    cnt = (C.c_uint64 * 3)()
    capi.check(lib.sa_align_profile_read(ms, cnt))
    lib.sa_align_profile(0)
    return elapsed, list(ms), [int(c) for c in cnt]


def roofline_of(batch, typ, cells_rank, steps, ms, cnt, pmc_key=None):
    """Roofline object of the dominant kernel (the DP fill) of one alignment workload, per launch."""
    fill_launches = max(cnt[1], 1)
    fill_ms = ms[1] / fill_launches
    cells_per_launch = cells_rank * steps / fill_launches
    fill_cups = cells_per_launch / (fill_ms * 1e-3)
    achieved_tops = fill_cups * OPS_PER_CELL[typ] / 1e12
    name = batch.kernel_name()
    peak_tops = PEAK_PK16_TOPS if "pk16" in name else PEAK_I32_TOPS
    r = {
        "kernel": name, "bound": "valu_int", "achieved": achieved_tops, "peak": peak_tops,
        "unit": "Tops/s (integer lane-ops; %d per cell update)" % OPS_PER_CELL[typ],
        "frac": achieved_tops / peak_tops, "traffic": None,
        # both integer peaks, whatever the kernel computes in (a 32-bit kernel cannot exceed half the packed one)
        "frac_of_int32_peak": achieved_tops / PEAK_I32_TOPS, "frac_of_pk16_peak": achieved_tops / PEAK_PK16_TOPS,
        "fill_ms_per_launch": fill_ms, "fill_gcups": fill_cups / 1e9,
        "launches_per_step": {"pack": cnt[0] // steps, "fill": cnt[1] // steps, "traceback": cnt[2] // steps},
        "ms_per_step_by_kernel": {"pack": ms[0] / steps, "fill": ms[1] / steps, "traceback": ms[2] / steps},
        # direction bits written by the fill: 2 bits per cell
        "hbm": {"bound": "hbm", "achieved": cells_per_launch / 4 / (fill_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": cells_per_launch / 4 / (fill_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "algorithmic_bytes": cells_per_launch / 4,
    }
    if pmc_key:
        tr = pmc_traffic("bench", pmc_key)
        if tr is not None:
            r["traffic"] = tr
            r["traffic_unit"] = ("bytes per fill launch, mean over the launches of a step (%s, collected at commit %s "
                                 "with the option no_overlap, which keeps a step's fill in one launch)"
                                 % (os.path.relpath(PMC_FILE, ROOT), pmc_commit()))
    return r


def device_pairs_semiglobal(first, count, seed, dev, read_len=10_000, window_len=11_000, lead=500, sub=0.04,
                            ins=0.03, dele=0.03, block=1000):
    """Pairs [first, first + count) of the semi-global ONT-like workload, generated on the device: target =
    uniform ACGT window, query = target[lead : lead + read_len] with ~10 % edits (per base: deleted, preceded
    by a random inserted base, or substituted), cut / padded with random bases to exactly read_len.  Pairs are
    generated in blocks of `block` from a generator seeded by the block number, so any rank that generates a
    given pair generates the same bytes.  -> (queries [count, read_len] uint8, targets [count, window_len])."""
    import torch
    bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    qs, ts = [], []
    b0 = first // block
    b1 = (first + count + block - 1) // block
    for b in range(b0, b1):
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1_000_003 + b)
        m = block
        tcode = torch.randint(0, 4, (m, window_len), generator=g, device=dev, dtype=torch.int64)
        src = tcode[:, lead:lead + read_len]
        u = torch.rand((m, read_len), generator=g, device=dev)
        deleted = u < dele
        inserted = (u >= dele) & (u < dele + ins)
        hit = torch.rand((m, read_len), generator=g, device=dev) < sub
        shift = torch.randint(1, 4, (m, read_len), generator=g, device=dev, dtype=torch.int64)
        body = torch.where(hit, (src + shift) % 4, src)
        counts = (~deleted).to(torch.int64) + inserted.to(torch.int64)
        flat_counts = counts.reshape(-1)
        out = torch.repeat_interleave(body.reshape(-1), flat_counts)
        starts = torch.cumsum(flat_counts, 0) - flat_counts
        ins_pos = starts[inserted.reshape(-1)]       # the first copy of an inserted base's pair is the random one
        out[ins_pos] = torch.randint(0, 4, (ins_pos.numel(),), generator=g, device=dev, dtype=torch.int64)
        row_len = counts.sum(1)
        row_start = torch.cumsum(row_len, 0) - row_len
        j = torch.arange(read_len, device=dev)[None, :]
        idx = (row_start[:, None] + j).clamp(max=out.numel() - 1)
        pad = torch.randint(0, 4, (m, read_len), generator=g, device=dev, dtype=torch.int64)
        qcode = torch.where(j < row_len[:, None], out[idx], pad)
        lo = max(first - b * block, 0)
        hi = min(first + count - b * block, block)
        qs.append(bases[qcode[lo:hi]])
        ts.append(bases[tcode[lo:hi]])
    return torch.cat(qs), torch.cat(ts)


def device_batch_uniform(q_dev, t_dev, typ, dev_index):
    """DeviceAlignBatch over uniform-length pairs that already lie in device memory ([n, L] uint8 tensors)."""
    import torch
    from sequencealigner_amd import align
    n, ql = q_dev.shape
    tl = t_dev.shape[1]
    slack = torch.zeros(8, dtype=torch.uint8, device=q_dev.device)
    d_q = torch.cat([q_dev.reshape(-1), slack])
    d_t = torch.cat([t_dev.reshape(-1), slack])
    d_qo = torch.arange(n + 1, dtype=torch.int64, device=q_dev.device) * ql
    d_to = torch.arange(n + 1, dtype=torch.int64, device=q_dev.device) * tl
    return align.DeviceAlignBatch.from_device(d_q, d_qo, d_t, d_to, np.full(n, ql), np.full(n, tl), typ, 1, -1, -1,
                                              device=dev_index)


def rescore_all(b, match=1, mismatch=-1, gap=-1, chunk=4096):
    """Every pair's run list re-scored on the device: (scores equal, query consumed min / max)."""
    import torch
    n, stride = b.n, b.stride
    ok = True
    qmin, qmax = 1 << 62, 0
    col = torch.arange(stride, device=b.device)[None, :]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        runs = b.d_runs[lo * stride:hi * stride].reshape(hi - lo, stride).to(torch.int64) & 0xFFFFFFFF
        valid = col >= (stride - b.d_nruns[lo:hi].to(torch.int64))[:, None]
        length = (runs >> 2) * valid
        op = runs & 3
        unit = torch.where(op == 0, match, torch.where(op == 1, mismatch, gap))
        ok &= bool(torch.equal((length * unit).sum(1), b.d_score[lo:hi].to(torch.int64)))
        qu = (length * (op != 3)).sum(1)
        qmin, qmax = min(qmin, int(qu.min())), max(qmax, int(qu.max()))
    return ok, qmin, qmax


def leg_nw_1k(lib, dev, steps, warmup, threads, with_cpu):
    from sequencealigner_amd import align
    import torch
    typ, n, ql, tl = WORKLOADS["nw_1kx1k"]
    q, t = make_pairs("nw_1kx1k", n, seed=1001)
    b = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, device=dev)
    elapsed, ms, cnt = timed_alignment(lib, b, steps, warmup)
    out = {"workload": f"nw_1kx1k: {n} pairs, query {ql} x target {tl}, Global, match 1 mismatch -1 gap -1, "
                       "full CIGAR + target_begin",
           "value": b.cells * steps / elapsed / 1e9, "unit": "GCUPS", "ms_per_step": elapsed / steps * 1e3,
           "steps": steps, "roofline": roofline_of(b, typ, float(b.cells), steps, ms, cnt, "fill_wave_pk16<0, 16, false>")}
    if with_cpu:
        cpu, cpu_scores = cpu_baseline(q, t, typ, n, threads)
        cpu["scores_equal_gpu"] = bool(np.array_equal(cpu_scores, b.scores()))
        out["cpu_baseline"] = cpu
    del b
    torch.cuda.empty_cache()
    return out


def leg_sg_10k(lib, dev, steps, warmup, threads, with_cpu, n, first=0, world=None, sdist=None, total=None):
    import torch
    typ = 2
    tdev = torch.device("cuda", dev)
    q_dev, t_dev = device_pairs_semiglobal(first, n, seed=3, dev=tdev)
    b = device_batch_uniform(q_dev, t_dev, typ, dev)
    elapsed, ms, cnt = timed_alignment(lib, b, steps, warmup, world, sdist)
    cells = float(b.cells)
    if world is not None and world.is_dist:
        elapsed = sdist.reduce_max(world, elapsed)
        cells_all = sdist.reduce_sum(world, cells)
    else:
        cells_all = cells
    ok, qmin, qmax = rescore_all(b)
    out = {"workload": (f"sg_10kx11k: {n} pairs on this GPU" +
                        (f" of a fixed set of {total} split over {world.size} GPUs" if total else "") +
                        ", query 10000 x target 11000 (~10 % edits), Semiglobal, match 1 mismatch -1 gap -1, "
                        "full CIGAR"),
           "scaling": "strong" if total else "weak",
           "value": cells_all * steps / elapsed / 1e9, "unit": "GCUPS", "ms_per_step": elapsed / steps * 1e3,
           "steps": steps, "passes_per_step": cnt[1] // steps,
           "workspace_GB": b.workspace_bytes / 1e9,
           "cigars_rescore_to_score": ok, "query_consumed_min_max": [qmin, qmax],
           "roofline": roofline_of(b, typ, cells, steps, ms, cnt, "fill_wave_pk16<2, ")}  # (the only leg with these kernels)
    if with_cpu:
        sample = min(n, max(threads, 16))
        q = q_dev[:sample].cpu().numpy()
        t = t_dev[:sample].cpu().numpy()
        cpu, cpu_scores = cpu_baseline(q, t, typ, sample, threads,
                                       " (880 MB of matrix per pair; the rate is extrapolated to the whole set)")
        cpu["scores_equal_gpu"] = bool(np.array_equal(cpu_scores, b.scores()[:sample]))
        out["cpu_baseline"] = cpu
    del b, q_dev, t_dev
    torch.cuda.empty_cache()
    return out


def leg_nw_100k(lib, dev, steps, warmup, with_cpu):
    import torch
    from sequencealigner_amd import align, synth
    typ, L = 0, 100_000
    qq, tt = synth.pair_global_long(L, seed=5)
    b = align.DeviceAlignBatch([qq], [tt], typ, 1, -1, -1, device=dev)
    elapsed, ms, cnt = timed_alignment(lib, b, steps, warmup)
    ok, qmin, qmax = rescore_all(b)
    score = int(b.scores()[0])
    out = {"workload": f"nw_100kx100k: 1 pair, query {L} x target {L} (~10 % edits), Global, match 1 mismatch -1 "
                       "gap -1, full CIGAR (2.5 GB of direction bits)",
           "value": b.cells * steps / elapsed / 1e9, "unit": "GCUPS", "ms_per_step": elapsed / steps * 1e3,
           "steps": steps, "score": score, "cigar_rescores_to_score": ok,
           "query_consumed": qmax, "roofline": roofline_of(b, typ, float(b.cells), steps, ms, cnt, "fill_wave_coop2<0")}
    if with_cpu:   # the full matrix would need 80 GB: the CPU checker's two-row scorer gives the score alone
        cpu_checker, _ = cpu_checker_lib()
        t0 = time.perf_counter()
        want = cpu_checker.restatement().score_linear(bytes(qq), bytes(tt), typ, 1, -1, -1)
        dt = time.perf_counter() - t0
        out["cpu_scorer"] = {"score": want, "score_equal_gpu": want == score, "seconds": dt, "cores": 1,
                             "GCUPS": b.cells / dt / 1e9,
                             "what": "linear-memory CPU scorer (score only, no traceback), one thread"}
    del b
    torch.cuda.empty_cache()
    return out


def pcie_leg(lib, q, t, typ, pairs):
    """PCIe-inclusive figures of the headline workload through the host-buffer entry point (never the
    headline value): the copies alone, and sa_align_batch_host end to end without / with CIGAR strings."""
    import torch
    from sequencealigner_amd import capi
    n = min(pairs, q.shape[0])
    q, t = np.ascontiguousarray(q[:n]), np.ascontiguousarray(t[:n])
    hq, ht = torch.from_numpy(q).pin_memory(), torch.from_numpy(t).pin_memory()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    dq, dt_ = hq.cuda(non_blocking=True), ht.cuda(non_blocking=True)   # warm
    torch.cuda.synchronize()
    ev[0].record()
    dq, dt_ = hq.cuda(non_blocking=True), ht.cuda(non_blocking=True)
    ev[1].record()
    res = torch.zeros(n * 3 + n * 4, dtype=torch.int32, device="cuda")   # score, target_begin, nruns + ~4 runs
    hres = torch.empty(res.shape, dtype=torch.int32).pin_memory()
    ev[2].record()
    hres.copy_(res, non_blocking=True)
    ev[3].record()
    torch.cuda.synchronize()
    qp = (C.c_char_p * n)(*[C.c_char_p(q.ctypes.data + i * q.shape[1]) for i in range(n)])
    tp = (C.c_char_p * n)(*[C.c_char_p(t.ctypes.data + i * t.shape[1]) for i in range(n)])
    ql = (C.c_uint32 * n)(*([q.shape[1]] * n))
    tl = (C.c_uint32 * n)(*([t.shape[1]] * n))
    scores, tb = (C.c_int32 * n)(), (C.c_uint32 * n)()
    best = {}
    for want_cigar in (False, True):
        best[want_cigar] = 1e9
        for _ in range(3):
            cig = (C.c_void_p * n)() if want_cigar else None
            t0 = time.perf_counter()
            capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, n, typ, 1, -1, -1, 1, scores, tb, cig))
            best[want_cigar] = min(best[want_cigar], time.perf_counter() - t0)
            if want_cigar:
                for i in range(n):
                    lib.sa_free(cig[i])
    cells = float(n) * q.shape[1] * t.shape[1]
    return {"pairs": n, "h2d_ms": ev[0].elapsed_time(ev[1]), "h2d_bytes": int(q.nbytes + t.nbytes),
            "d2h_ms": ev[2].elapsed_time(ev[3]), "d2h_bytes": int(res.numel() * 4),
            "host_call_ms_scores_only": best[False] * 1e3, "host_call_ms_with_cigar_text": best[True] * 1e3,
            "cigar_text_ms": (best[True] - best[False]) * 1e3,
            "gcups_pcie_inclusive_scores_only": cells / best[False] / 1e9,
            "gcups_pcie_inclusive_with_cigar_text": cells / best[True] / 1e9,
            "note": "sa_align_batch_host: pinned staging + H2D + kernels + D2H (+ one malloc'ed CIGAR string per pair)"}


def minimizer_leg(dev, steps, cpu_threads, with_cpu, only=None):
    """Secondary metric of the same hot path: MinimizeBinary k=15 w=5 on a 4.6 Mbp genome (one sequence)
    and on 1 Gbp of 10 kbp reads (100 000 sequences), inputs resident in HBM.  Returns a dict."""
    import torch
    from sequencealigner_amd import minimize, synth
    out = {"k": 15, "w": 5}
    for name, make in (("genome_4.6Mbp", lambda: [synth.genome(4_600_000, seed=4)]),
                       ("reads_100k_x_10kbp", lambda: synth.reads(100_000, 10_000, seed=6))):
        if only and not name.startswith(only):
            continue
        seqs = make()
        section = "minimizers_reads" if name.startswith("reads") else "minimizers_genome"
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
        # HBM bytes of the leg's kernels from the counter passes over this leg alone
        kernels = {}
        for e_name in ("min_run<sa::Machine<5u, false>, 15u>", "sort_emit_small", "bucket_scatter", "bucket_plan"):
            tr = pmc_traffic(section, e_name)
            if tr is not None:
                es = pmc_entries(section, e_name)
                kernels[e_name.split("<")[0]] = {"traffic": tr, "fetch_bytes": 2048.0 * es[0]["FETCH_SIZE"],
                                                 "write_bytes": 1024.0 * es[0]["WRITE_SIZE"],
                                                 "valu_wave_instructions": es[0].get("SQ_INSTS_VALU")}
        traffic = sum(k["traffic"] for k in kernels.values()) if kernels else None
        tops = b.windows / dt * ops / 1e12
        gbs = bytes_alg / dt / 1e9
        out[name] = {"ms": dt * 1e3, "windows_per_s": b.windows / dt, "minimizers_per_s": n_out / dt,
                     "unique_minimizers": n_out,
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": bytes_alg,
                                  "traffic": traffic, "traffic_by_kernel": kernels,
                                  "traffic_unit": "bytes per call, summed over its kernels (%s, commit %s)"
                                                  % (os.path.relpath(PMC_FILE, ROOT), pmc_commit()),
                                  "fetch_over_bases": (kernels["min_run"]["fetch_bytes"] / b.total_len
                                                       if "min_run" in kernels else None),
                                  "valu_int": {"achieved": tops, "peak": PEAK_I32_TOPS,
                                               "unit": "Tops/s (integer lane-ops; %d per window)" % ops,
                                               "frac": tops / PEAK_I32_TOPS}}}
        if with_cpu and name.startswith("reads"):
            _, chk = cpu_checker_lib()
            sample = min(len(seqs), 100_000)
            blob = np.concatenate([np.ascontiguousarray(seqs[:sample]).reshape(sample, -1),
                                   np.zeros((sample, 1), np.uint8)], axis=1)   # NUL-terminated
            off = np.arange(sample, dtype=np.uint64) * np.uint64(blob.shape[1])
            lens = np.full(sample, seqs.shape[1], dtype=np.uint32)
            t0 = time.perf_counter()
            counts, hashes = chk.minimize_batch(blob.reshape(-1), off, lens, 15, 5, cpu_threads)
            cdt = time.perf_counter() - t0
            cpu_checker, _ = cpu_checker_lib()
            gpu_hashes = cpu_checker.triple_list_hashes(b.d_out, b.d_out_off)[:sample]
            win = sample * (seqs.shape[1] - 18)
            out[name]["cpu_baseline"] = {
                "value": win / cdt, "unit": "windows/s", "cores": cpu_threads, "cores_on_box": os.cpu_count(),
                "kind": "reference" if chk.kind == "reference" else "port",
                "sample": f"{sample} reads of the same workload, {cdt:.1f} s",
                # every read's list: equal length and equal hash over its (value, position, strand) entries
                "triples_equal_gpu": bool(np.array_equal(counts.astype(np.int64), b.counts()[:sample]) and
                                          np.array_equal(hashes, gpu_hashes))}
        del b
        torch.cuda.empty_cache()
    return out


def sg_split(total, world_size, rank):
    """The semi-global leg under N > 1 ranks: one fixed set of `total` pairs of equal cost, dealt by cost like the
    library's own shards (sequencealigner_amd.dist.shard_by_cost): equal costs give contiguous blocks.
    -> (first pair of this rank, pairs of this rank)"""
    from sequencealigner_amd import dist as sdist
    shards = sdist.shard_by_cost(np.full(total, 10_000 * 11_000, dtype=np.int64), world_size)
    mine = shards[rank]
    return (int(mine[0]) if len(mine) else 0), len(mine)


def host_queue_leg(lib, pairs_per_device):
    """sa_align_batch_host - the library's own queue: one host thread, stream and set of buffers per GPU - over
    every GPU this process sees, on semi-global 10 kbp x 11 kbp pairs (21 KB of input per 1.1e8 cells: the shape
    on which a node's GPUs are not held back by the host's staging copies).  Wall time of the call, PCIe and
    CIGAR text included, and the time every device's shard spent in its stages (the library's SA_HOST_TRACE
    lines, read back from stderr)."""
    import tempfile
    import torch
    from sequencealigner_amd import capi
    ndev = int(lib.sa_device_count())
    n = pairs_per_device * ndev
    q_dev, t_dev = device_pairs_semiglobal(0, n, seed=11, dev=torch.device("cuda", torch.cuda.current_device()))
    q, t = q_dev.cpu().numpy(), t_dev.cpu().numpy()
    del q_dev, t_dev
    qp = (C.c_char_p * n)(*[C.c_char_p(q.ctypes.data + i * q.shape[1]) for i in range(n)])
    tp = (C.c_char_p * n)(*[C.c_char_p(t.ctypes.data + i * t.shape[1]) for i in range(n)])
    ql = (C.c_uint32 * n)(*([q.shape[1]] * n))
    tl = (C.c_uint32 * n)(*([t.shape[1]] * n))
    scores, tb = (C.c_int32 * n)(), (C.c_uint32 * n)()

    def call():
        cig = (C.c_void_p * n)()
        t0 = time.perf_counter()
        capi.check(lib.sa_align_batch_host(qp, ql, tp, tl, n, 2, 1, -1, -1, 0, scores, tb, cig))
        dt = time.perf_counter() - t0
        for i in range(n):
            lib.sa_free(cig[i])
        return dt

    call()                                   # warm: buffers, streams, code objects on every device
    best = min(call() for _ in range(2))
    # once more with the stage trace on (the library reads SA_HOST_TRACE at every shard), stderr into a file
    per_device = {}
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        sys.stderr.flush()
        saved = os.dup(2)
        os.environ["SA_HOST_TRACE"] = "1"
        try:
            os.dup2(tmp.fileno(), 2)
            traced = call()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            del os.environ["SA_HOST_TRACE"]
        tmp.seek(0)
        for line in tmp.read().decode(errors="replace").splitlines():
            f = line.split()
            if len(f) >= 6 and f[0] == "[sa" and f[1] == "host" and f[2] == "dev":
                d = f[3].rstrip("]")
                per_device[d] = per_device.get(d, 0.0) + float(f[-2])
    cells = float(n) * q.shape[1] * t.shape[1]
    return {"workload": f"host_queue: {n} pairs ({pairs_per_device} per GPU), query 10000 x target 11000, Semiglobal, "
                        "host buffers in, CIGAR strings out, one call of sa_align_batch_host over all GPUs",
            "devices": ndev, "value": cells / best / 1e9, "unit": "GCUPS (PCIe and CIGAR text included)",
            "ms_per_call": best * 1e3, "ms_per_call_traced": traced * 1e3,
            "device_busy_ms": per_device}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sw_150x300", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: the workload's size)")
    ap.add_argument("--no-cigar", action="store_true", help="score only (not the headline number)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="pairs for the CPU baseline (0 = skip all CPU legs)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-minimizers", action="store_true", help="skip the secondary minimizer metric")
    ap.add_argument("--no-legs", action="store_true", help="skip the other alignment shapes")
    ap.add_argument("--legs", default="nw_1kx1k,sg_10kx11k,nw_100kx100k,pcie,host_queue",
                    help="comma-separated legs to run besides the headline")
    ap.add_argument("--leg-steps", type=int, default=0, help="timed steps of every leg (default: 5 / 2 / 3)")
    ap.add_argument("--sg-pairs", type=int, default=25_000, help="semi-global leg: pairs per GPU (one GPU)")
    ap.add_argument("--sg-total", type=int, default=200_000,
                    help="semi-global leg under --gpus N > 1: size of the fixed set that is split over the ranks")
    ap.add_argument("--queue-pairs", type=int, default=1024, help="host_queue leg: pairs per GPU")
    ap.add_argument("--only-minimizers", default="", help="run only this minimizer leg (reads | genome): counter passes")
    ap.add_argument("--split-only", action="store_true",
                    help="no GPU work: every rank computes its share of the semi-global set, rank 0 prints the line")
    args = ap.parse_args()

    if args.split_only:   # the N-rank split logic, runnable on CPU ranks (gloo)
        from sequencealigner_amd import dist as sdist
        world = sdist.init()
        first, count = sg_split(args.sg_total, world.size, world.rank)
        total = sdist.reduce_sum(world, count)
        last = sdist.reduce_max(world, first + count)
        out = None
        if world.rank == 0:
            out = {"metric": "split only", "n_gpus": world.size,
                   "legs": {"sg_10kx11k": {"scaling": "strong" if world.is_dist else "weak",
                                           "pairs_this_rank": count, "first_pair_this_rank": first,
                                           "pairs_all_ranks": int(total), "end_of_last_shard": int(last),
                                           "sg_total": args.sg_total}}}
            print(json.dumps(out), flush=True)
        sdist.finalize(world)
        return out
    if args.only_minimizers:
        import torch
        from sequencealigner_amd import capi
        torch.cuda.set_device(0)
        capi.lib()
        out = {"minimizers": minimizer_leg(0, 1, 1, False, only=args.only_minimizers)}
        print(json.dumps(out), flush=True)
        return out

    import torch
    from sequencealigner_amd import align, capi, dist as sdist

    world = sdist.init()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    dev = sdist.device_of(world)
    torch.cuda.set_device(dev)
    lib = capi.lib()
    threads = args.cpu_threads or host_cores()[0]
    legs_wanted = [] if args.no_legs else [x for x in args.legs.split(",") if x]

    typ, default_pairs, ql, tl = WORKLOADS[args.workload]
    n = args.pairs or default_pairs
    q, t = make_pairs(args.workload, n, seed=1000 + world.rank)
    batch = align.DeviceAlignBatch(q, t, typ, 1, -1, -1, want_cigar=not args.no_cigar,
                                   want_target_begin=True, device=dev)
    elapsed, ms, cnt = timed_alignment(lib, batch, args.steps, args.warmup, world, sdist)

    elapsed_max = sdist.reduce_max(world, elapsed)
    cells_rank = float(batch.cells)
    cells_all = sdist.reduce_sum(world, cells_rank)
    value = cells_all * args.steps / elapsed_max / 1e9
    full = args.workload == "sw_150x300" and n == default_pairs and not args.no_cigar
    roofline = roofline_of(batch, typ, cells_rank, args.steps, ms, cnt,
                           "fill_lane_pk16<1" if full else None)
    kernel = batch.kernel_name()
    gpu_scores = batch.scores() if world.rank == 0 else None
    del batch
    torch.cuda.empty_cache()

    # the semi-global leg under N > 1 GPUs: one fixed set, split over the ranks (strong scaling)
    sg_strong = None
    if world.is_dist and "sg_10kx11k" in legs_wanted:
        first, count = sg_split(args.sg_total, world.size, world.rank)   # equal costs: a contiguous block
        sg_strong = leg_sg_10k(lib, dev, 1, 1, threads, False, count, first=first, world=world,
                               sdist=sdist, total=args.sg_total)
    sdist.barrier(world)   # every rank's GPU work is done: what follows on rank 0 has the node to itself

    out = None
    if world.rank == 0:
        cpu = None
        sample = args.cpu_sample
        if sample < 0:
            sample = min(n, 1_000_000 if args.workload == "sw_150x300" else 300)
        with_cpu = sample > 0 and not world.is_dist
        if world.is_dist and sample > 0:
            # N > 1: a bounded sample on rank 0's share of the host cores, after the timed region (the other
            # ranks wait in the barrier at the end)
            sample = min(sample, 20_000)
            cpu, cpu_scores = cpu_baseline(q, t, typ, sample, threads,
                                           f" (rank 0 of {world.size}, after the timed region)")
            cpu["scores_equal_gpu"] = bool(np.array_equal(cpu_scores, gpu_scores[:sample]))
        elif with_cpu:
            cpu, cpu_scores = cpu_baseline(q, t, typ, sample, threads)
            cpu["scores_equal_gpu"] = bool(np.array_equal(cpu_scores, gpu_scores[:sample]))
        out = {
            "metric": "GCUPS (DP cell updates/s), batched Align() incl. traceback + CIGAR",
            "value": value, "unit": "GCUPS", "n_gpus": world.size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16x2" if "pk16" in kernel else "int32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} pairs/GPU, query {ql} x target {tl}, "
                                   f"{TYPE_NAME[typ]}, match 1 mismatch -1 gap -1, "
                                   f"{'score only' if args.no_cigar else 'full CIGAR + target_begin'}",
                       "pairs_per_gpu": n, "cells_per_step_per_gpu": cells_rank},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        legs = {}
        if sg_strong is not None:
            legs["sg_10kx11k"] = sg_strong
        # (under N > 1 ranks these run on rank 0's GPU after the timed region: per-GPU figures, labelled as such)
        if "pcie" in legs_wanted and args.workload == "sw_150x300" and not args.no_cigar:
            out["pcie"] = pcie_leg(lib, q, t, typ, 200_000)
        if "nw_1kx1k" in legs_wanted:
            legs["nw_1kx1k"] = leg_nw_1k(lib, dev, args.leg_steps or 5, 2 if not args.leg_steps else 1, threads,
                                         with_cpu)
        if "sg_10kx11k" in legs_wanted and not world.is_dist:
            legs["sg_10kx11k"] = leg_sg_10k(lib, dev, args.leg_steps or 2, 1, threads, with_cpu, args.sg_pairs)
        if "nw_100kx100k" in legs_wanted:
            legs["nw_100kx100k"] = leg_nw_100k(lib, dev, args.leg_steps or 3, 1, with_cpu)
        if world.is_dist:
            for name in ("nw_1kx1k", "nw_100kx100k"):
                if name in legs:
                    legs[name]["n_gpus"] = 1
        if "host_queue" in legs_wanted:   # one process, every visible GPU (the other ranks are idle by now)
            legs["host_queue"] = host_queue_leg(lib, args.queue_pairs)
        if legs:
            out["legs"] = legs
        if not args.no_minimizers:
            out["minimizers"] = minimizer_leg(dev, args.leg_steps or 3, threads, with_cpu)
        print(json.dumps(out), flush=True)
    sdist.finalize(world)
    return out


if __name__ == "__main__":
    main()
