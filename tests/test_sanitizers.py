"""AddressSanitizer + UBSan over the host-side code that can run without a GPU: the product's planner
recipes (kofft_amd/csrc/tables.cpp) and the C oracle.  GPU sanitizers are not available on the device pool."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_tables_and_oracle_under_asan_ubsan(tmp_path):
    exe = tmp_path / "sanitize_tables"
    flags = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1",
             "-ffp-contract=off"]
    obj = tmp_path / "oracle.o"
    subprocess.run(["gcc", "-std=c11", "-c", *flags, str(ROOT / "oracle" / "kofft_oracle.c"), "-o", str(obj)], check=True,
                   capture_output=True, text=True)
    subprocess.run(["g++", "-std=c++17", *flags, str(ROOT / "tests" / "cpp" / "sanitize_tables.cpp"),
                    str(ROOT / "kofft_amd" / "csrc" / "tables.cpp"), str(obj), "-lm", "-o", str(exe)], check=True,
                   capture_output=True, text=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 problems" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
