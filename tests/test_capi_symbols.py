"""CPU suite: the C-ABI library loads and exports every function include/sa_hip.h declares, and the C++
drop-in exports the reference's own symbols.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

from sequencealigner_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sequencealigner_amd", "csrc")


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 14
    lib = capi.lib()
    bound = {n for n, _, _ in capi.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), n
        assert n in bound, f"{n} is declared in sa_hip.h but has no ctypes prototype in capi.SYMBOLS"
    assert bound <= set(names)


def test_error_reporting_without_gpu_calls():
    lib = capi.lib()
    # argument validation happens before any HIP call
    b = capi.AlignBatch()
    b.type = 9
    rc = lib.sa_align_batch_device(0, None, ctypes.byref(b), None, 0)
    assert rc == capi.SA_ERR_ARGUMENT
    assert b"unknown alignment type" in lib.sa_last_error()
    m = capi.MinimizeBatch()
    m.n = 1
    m.k, m.w = 15, 40
    m.d_seq = m.d_off = m.d_out = m.d_out_off = 8
    rc = lib.sa_minimize_batch_device(0, None, ctypes.byref(m), None, 0)
    assert rc == capi.SA_ERR_ARGUMENT


def test_alignment_planning_is_host_only():
    lib = capi.lib()
    b = capi.AlignBatch()
    b.n, b.max_qlen, b.max_tlen, b.min_qlen, b.min_tlen = 1_000_000, 150, 300, 150, 300
    b.type, b.match, b.mismatch, b.gap = 1, 1, -1, -1
    mn = ctypes.c_size_t(0)
    full = lib.sa_align_workspace_bytes(ctypes.byref(b), ctypes.byref(mn))
    assert 0 < mn.value < full < 64 << 30
    assert lib.sa_align_kernel_name(ctypes.byref(b)) == b"lane_pk16"
    b.match = 1000   # scores leave the 16-bit window: the 32-bit kernel is chosen
    assert lib.sa_align_kernel_name(ctypes.byref(b)) == b"lane_i32"
    # (the minimizer workspace size includes rocprim's temporary storage, which rocprim derives from the
    # device's architecture: that query needs a GPU and is covered by the GPU tests)
    assert lib.sa_minimize_out_capacity(2, 1000, 15, 5) == 1000 - 2 * 18


def test_dropin_exports_reference_symbols():
    so = os.path.join(CSRC, "libsequencealigner_amd.so")
    assert os.path.exists(so)
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", so], capture_output=True, text=True,
                         check=True).stdout
    for sig in ("Align(char const*, unsigned int, char const*, unsigned int, AlignmentType, int, int, int, "
                "std::__cxx11::basic_string", "NeedlemanWunschAlgorithm(char const*", "SmithWatermanAlgorithm(",
                "SemiGlobalAlgorithm(", "CompressCigar(std::__cxx11::basic_string", "MinimizeBinary(char const*, "
                "unsigned int, unsigned int, unsigned int)", "Minimize(char const*, unsigned int, unsigned int, "
                "unsigned int)", "removeDuplicate(std::vector<std::tuple<unsigned int, unsigned int, bool>"):
        assert sig in out, sig
