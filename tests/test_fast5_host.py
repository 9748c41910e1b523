"""CPU check of the register-level MinimizeBinary machine of the HIP kernels (sa::Fast5,
sequencealigner_amd/csrc/minimizer_fast.h): tests/cpp/test_fast5.cpp steps its host build beside a plain
restatement of the reference's machine (src/minimizer_binary.cpp:99-133, 149-172) and compares every window."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast5_matches_the_reference_machine_window_by_window(tmp_path):
    exe = str(tmp_path / "test_fast5")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "sequencealigner_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_fast5.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "fast5 ok" in out, out
