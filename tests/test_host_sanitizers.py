"""CPU-only: the host graph builder and the bincode (de)serialiser under AddressSanitizer + UndefinedBehaviorSanitizer
(SURVEY.md §5 suggested it; VERDICT round 2 housekeeping).  `build_host.sh sanitize` compiles cleora_host.cpp with
-fsanitize=address,undefined into libcleora_host_san.so; the property test of tests/test_builder_hypothesis.py (arbitrary
small hypergraphs against the Python builder oracle, bincode round trips) and the corruption fuzz of
tests/test_wire_and_generator.py then run in a subprocess that loads THAT library (CLEORA_HOST_LIB) with libasan
preloaded.  Any heap overflow, use after free, misaligned access or signed overflow in the builder aborts the subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_builder_and_wire_format_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan is not installed with this gcc")
    subprocess.check_call(["bash", os.path.join(ROOT, "cleora_amd", "csrc", "build_host.sh"), "sanitize"])
    env = dict(os.environ)
    env.update(CLEORA_HOST_LIB=os.path.join(ROOT, "cleora_amd", "libcleora_host_san.so"), LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_builder_hypothesis.py",
                        "tests/test_wire_and_generator.py", "-k", "not bench_c5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
