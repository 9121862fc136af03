"""AddressSanitizer + UndefinedBehaviorSanitizer run of the kernels' math headers (host build of the
same source the GPU compiles; GPU sanitizers are unavailable on the pool) with finite-difference
checks of every hand-derived backward."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_math_headers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selftest")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "cpu_harness", "selftest.cpp")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "selftest: ok" in out.stdout
