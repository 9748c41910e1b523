"""GPU test of sa_align_batch_device_static, the alignment batch call that allocates nothing (its scratch
memory is a static array inside the library): same scores, target_begin and CIGAR runs as
sa_align_batch_device, on consecutive calls that share the array, the later ones with smaller inputs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _outputs(b):
    import torch
    torch.cuda.synchronize()
    runs = b.d_runs.cpu().numpy().view(np.uint32).reshape(b.n, b.stride)
    nruns = b.d_nruns[:b.n].cpu().numpy()
    valid = np.arange(b.stride)[None, :] >= (b.stride - nruns)[:, None]
    return b.scores().copy(), b.target_begins().copy(), nruns.copy(), np.where(valid, runs, 0)


def _both_ways(b):
    """Runs the batch through the workspace entry point, clears the outputs, runs it through the static
    entry point and returns both sets of outputs."""
    b.run()
    want = _outputs(b)
    for t in (b.d_score, b.d_tb, b.d_runs, b.d_nruns):
        t.fill_(-1)
    b.run_static()
    return want, _outputs(b)


def test_static_scratch_entry_point_equals_workspace_entry_point(oracle):
    import torch
    from sequencealigner_amd import align, capi, synth
    lib = capi.lib()
    scratch = lib.sa_align_static_scratch_bytes()
    assert scratch == 64 << 20
    # call 1: 16 000 Local pairs of the benchmark shape - the packed lane kernel, more than one pass of the array
    q, t = synth.pairs_short_read_window(16_000, seed=31)
    big = align.DeviceAlignBatch(q, t, 1, 1, -1, -1, workspace_bytes=1 << 30)
    assert big.kernel_name() == "lane_pk16"
    mn = C.c_size_t(0)
    assert lib.sa_align_workspace_bytes(C.byref(big.batch), C.byref(mn)) > scratch > mn.value
    want, got = _both_ways(big)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    for i in range(0, 16_000, 1601):   # and the workspace path is itself checked against the CPU checker
        assert (int(want[0][i]), big.cigars([i])[0], int(want[1][i])) == oracle.align(bytes(q[i]), bytes(t[i]), 1, 1, -1, -1)
    # call 2, right after it on the same array, smaller: 60 Global pairs.  The workspace entry point runs so few
    # pairs on the cooperative kernel, the static-scratch one never does (it only enqueues work) and takes the
    # packed wavefront kernel: two different kernels must agree
    q2, t2 = synth.pairs_global_1k(60, seed=32, length=700)
    small = align.DeviceAlignBatch(q2, t2[:, :650], 0, 1, -1, -1)
    assert small.kernel_name() == "wave_i32_coop"
    capi.set_option("force_coop", 0)
    try:
        assert small.kernel_name() == "wave_pk16"
    finally:
        capi.set_option("force_coop", -1)
    want, got = _both_ways(small)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    # call 3: 9 ragged Semiglobal pairs with a scoring the packed kernels do not take (32-bit kernels), then
    # call 1's batch again: nothing of the smaller calls may be left behind in the array
    rng = np.random.default_rng(5)
    qs = [synth.random_bases(rng, int(n)) for n in rng.integers(5, 900, size=9)]
    ts = [np.concatenate([synth.random_bases(rng, 30), x, synth.random_bases(rng, 20)]) for x in qs]
    odd = align.DeviceAlignBatch(qs, ts, 2, 3, 1, 2)
    assert "pk16" not in odd.kernel_name()
    want, got = _both_ways(odd)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    want, got = _both_ways(big)
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    # a batch whose single pass exceeds the array is refused with a clear error, nothing is launched
    qq, tt = synth.pair_global_long(20_000, seed=3)
    huge = align.DeviceAlignBatch([qq], [tt], 0, 1, -1, -1, workspace_bytes=1)
    rc = lib.sa_align_batch_device_static(0, None, C.byref(huge.batch))
    assert rc == capi.SA_ERR_WORKSPACE
    assert b"static array holds" in lib.sa_last_error()
    torch.cuda.synchronize()


def test_static_entry_point_allocates_nothing_and_can_be_captured(oracle):
    """What the entry point is for: device memory in use is the same before and after its calls, and - after
    sa_align_static_init, outside the capture - a call recorded into a HIP graph gives the eager result on every
    replay, with the program's own work and synchronisations between the replays."""
    import torch
    from sequencealigner_amd import align, capi, synth
    lib = capi.lib()
    capi.check(lib.sa_align_static_init(0))
    q, t = synth.pairs_short_read_window(6000, seed=33)
    b = align.DeviceAlignBatch(q, t, 1, 1, -1, -1)
    assert lib.sa_align_may_synchronize(0, C.byref(b.batch), b.workspace_bytes) == 0
    b.run()
    want = _outputs(b)
    b.run_static()      # warm: the kernels' code is resident from here on
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(3):
        b.run_static()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 == free0
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph, stream=side):
        b.run_static(torch.cuda.current_stream())
    torch.cuda.synchronize()
    for replay in range(3):
        for x in (b.d_score, b.d_tb, b.d_runs, b.d_nruns):
            x.fill_(-1)
        torch.cuda.synchronize()
        graph.replay()
        got = _outputs(b)
        for w, g in zip(want, got):
            assert np.array_equal(w, g), replay
        # eager work of the program between the replays
        other = align.DeviceAlignBatch(q[:500], t[:500], 2, 1, -1, -1)
        other.run_static()
        torch.cuda.synchronize()
    for i in range(0, 6000, 599):
        assert (int(want[0][i]), b.cigars([i])[0], int(want[1][i])) == oracle.align(bytes(q[i]), bytes(t[i]), 1, 1, -1, -1)
