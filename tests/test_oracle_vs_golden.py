"""CPU suite: the restatement in oracle/oracle.cpp against (1) the reference's own unit-test answers
(reference test/testAlignmentMinimizer.cpp:7-82) and (2) the fixtures produced by the compiled
reference (tests/golden/)."""
import hashlib

import numpy as np

import golden_io


def test_reference_unit_test_answers(oracle):
    # test/testAlignmentMinimizer.cpp:14,24,34 assert the scores only
    assert oracle.align("CTCTGTTCG", "CGTATCTTGA", 0, 0, -1, -1)[0] == -5
    assert oracle.align("CTCTGAG", "TGTCAGT", 1, 2, -2, -1)[0] == 6
    assert oracle.align("AGCATGCAAT", "ATCCGAACATCCAATCGAAGC", 2, 2, -1, -1)[0] == 14
    # :42-56 and :66-80
    assert oracle.minimize_binary("TGACGTACATGGACA", 3, 4) == [(2, 10, 0), (6, 4, 0), (6, 7, 1), (11, 1, 0)]
    assert oracle.minimize_decimal("TGACGTACATGGACA", 3, 4) == [(2, 10, 0), (12, 4, 0), (12, 7, 1), (23, 1, 0)]


def test_unknown_alignment_type(oracle):
    assert oracle.align("ACGT", "ACGT", 7, 1, -1, -1)[0] == 9999  # alignment.cpp:371-373


def _check_align(oracle, cases):
    for c in cases:
        got = oracle.align(c["q"], c["t"], c["type"], c["match"], c["mismatch"], c["gap"])
        assert got == (c["score"], c["cigar"], c["target_begin"]), c


def test_align_cases(oracle):
    cases = golden_io.load("align_cases.json")
    assert len(cases) > 500
    _check_align(oracle, cases)


def test_align_benchmark_shapes(oracle):
    _check_align(oracle, golden_io.load("align_shapes.json"))


def test_align_score_without_cigar(oracle):
    for c in golden_io.load("align_cases.json")[:200]:
        got = oracle.align(c["q"], c["t"], c["type"], c["match"], c["mismatch"], c["gap"],
                           want_cigar=False, want_target_begin=(c["type"] == 1))
        assert got[0] == c["score"]
        if c["type"] == 1:
            assert got[2] == c["target_begin"]


def test_minimizer_cases(oracle):
    cases = golden_io.load("minimizer_cases.json")
    assert len(cases) > 300
    for c in cases:
        fn = oracle.minimize_binary if c["variant"] == "binary" else oracle.minimize_decimal
        got = fn(c["seq"], c["k"], c["w"])
        want = None if c["out"] == "throws" else [tuple(x) for x in c["out"]]
        assert got == want, (c["seq"], c["k"], c["w"], c["variant"])


def test_minimizer_long(oracle):
    for c in golden_io.load("minimizer_long.json"):
        arr = oracle.minimize_binary_array(golden_io.long_sequence(c["name"]), c["k"], c["w"])
        assert arr.shape[0] == c["count"]
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == c["sha256"]
        assert arr[:8].tolist() == c["head"] and arr[-8:].tolist() == c["tail"]


def test_remove_duplicate(oracle):
    rng = np.random.default_rng(5)
    t = np.stack([rng.integers(0, 6, 200), rng.integers(0, 4, 200), rng.integers(0, 2, 200)], axis=1)
    sorted_in_place, uniq = oracle.remove_duplicate(t)
    want_sorted = sorted(tuple(int(v) for v in r) for r in t)
    assert sorted_in_place == want_sorted          # utility.cpp:11 sorts the caller's vector
    assert uniq == sorted(set(want_sorted))        # utility.cpp:16-24
