"""The host code of librt355 that parses untrusted files or indexes by computed offsets — the OpenEXR reader (rt_exr.cpp: PIZ Huffman
and wavelet, ZIP, RLE, hostile headers) and the host twins of the sphere grid, the direction-mask table and the candidate lists —
under AddressSanitizer and UndefinedBehaviorSanitizer: `make -C rsraytracingv3_amd/csrc host-asan` builds that code
alone with plain g++ (no GPU code, no HIP runtime calls), and the CPU-tier tests of those parts run against it in a child process
with libasan preloaded.  Any report aborts the child (-fno-sanitize-recover, halt_on_error)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rsraytracingv3_amd", "csrc")
LIB = os.path.join(ROOT, "rsraytracingv3_amd", "librt355_host_asan.so")
TESTS = ["tests/test_exr_loader.py", "tests/test_grid_host.py", "tests/test_dirmask_host.py", "tests/test_dirlist_host.py"]


def test_host_code_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", CSRC, "host-asan"], check=True, capture_output=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "no libasan next to gcc"
    env = dict(os.environ, LD_PRELOAD=libasan, RT355_LIB=LIB, RT355_HOST_ONLY="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + TESTS, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert "passed" in run.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
