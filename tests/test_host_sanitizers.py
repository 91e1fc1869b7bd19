"""Host-side sanitizer pass (CPU box only; SURVEY.md appendix: "validate index ranges on the host before launch" — GPU AddressSanitizer
and xnack+ code objects are not available on this pool, so the DEVICE half is covered by the parity tests and the HOST half by this).

`tools/build_sanitized.sh` builds libatx.so and the test-only RCCL stand-in with -fsanitize=address,undefined on the host compilation
(-Xarch_host); `tests/c_abi/sanitize_check.c`, built with the same flags, then drives every entry point's argument validation, the
host-side table builder over 176 program shapes with exactly-sized buffers, and the dlopen'ed collective binding.  Any report aborts.

What the first run found (round 4): the stand-in defined the ten NCCL entry points with `void*`-typed signatures and libatx called them
through typed pointers (UBSan -fsanitize=function) — both now share csrc/atx_nccl_abi.h, NCCL's own public types.  Nothing else:
no out-of-bounds access, no leak, no undefined arithmetic in the host paths."""

from __future__ import annotations

import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
HIPCC = "/opt/rocm/bin/hipcc"
VARIANTS = os.path.join(ROOT, "gpurun_out", "hostsan")  # scratch: git-ignored, outside the snapshot gpurun sends to the GPU box
LIB = os.path.join(VARIANTS, "libatx_hostsan.so")
STUB = os.path.join(VARIANTS, "librccl_stub_hostsan.so")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared-libasan", "-g"]


def asan_runtime_dir() -> str | None:
    found = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return os.path.dirname(found[0]) if found else None


@pytest.fixture(scope="module")
def sanitized():
    runtime = asan_runtime_dir()
    if not (os.path.exists(CLANG) and os.path.exists(HIPCC) and runtime):
        pytest.skip("clang / hipcc / the AddressSanitizer runtime are not installed")
    if not os.path.exists(os.path.join(ROOT, "tools", "build_sanitized.sh")):
        pytest.skip("the sanitizer recipe does not travel to the GPU box (.gpurunignore): this pass runs on the CPU box only")
    sources = (glob.glob(os.path.join(ROOT, "anemoi-transform_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
               [os.path.join(ROOT, "tests", "rccl_stub", "rccl_stub.cpp"), os.path.join(ROOT, "tools", "build_sanitized.sh")])
    newest = max(os.path.getmtime(p) for p in sources)
    if not (os.path.exists(LIB) and os.path.exists(STUB)) or min(os.path.getmtime(LIB), os.path.getmtime(STUB)) < newest:
        build = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_sanitized.sh")], capture_output=True, text=True, timeout=900)
        assert build.returncode == 0, build.stderr[-3000:]
    return runtime


def build_harness(tmp_path, runtime) -> str:
    exe = str(tmp_path / "sanitize_check")
    build = subprocess.run([CLANG, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", *SAN, "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "c_abi", "sanitize_check.c"), "-o", exe, "-L", VARIANTS, "-latx_hostsan",
                            f"-Wl,-rpath,{VARIANTS}", "-Wl,-rpath,/opt/rocm/lib", f"-Wl,-rpath,{runtime}"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    return exe


def test_host_paths_are_clean_under_asan_and_ubsan(sanitized, tmp_path):
    exe = build_harness(tmp_path, sanitized)
    env = dict(os.environ, ATX_RCCL_LIBRARY=STUB, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout[-2000:] + run.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
    assert "vector programs: checksum" in run.stdout


def test_the_pass_is_armed(sanitized, tmp_path):
    """Negative control: a caller that lies about the capacity of `out` makes the instrumented atx_vector_program write past a heap
    block — AddressSanitizer must stop the program inside libatx."""
    exe = build_harness(tmp_path, sanitized)
    run = subprocess.run([exe, "--lie-about-capacity"], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "heap-buffer-overflow" in run.stderr and "atx_vector_program" in run.stderr, run.stdout + run.stderr[-3000:]
    assert "not caught" not in run.stdout


def test_header_for_the_collective_abi_is_shared_with_the_stand_in():
    """libatx and the RCCL stand-in take NCCL's types and signatures from ONE header (what the UBSan function check asked for)."""
    comm = open(os.path.join(ROOT, "anemoi-transform_amd", "csrc", "atx_comm.hip")).read()
    stub = open(os.path.join(ROOT, "tests", "rccl_stub", "rccl_stub.cpp")).read()
    assert '#include "atx_nccl_abi.h"' in comm and "atx_nccl_abi.h" in stub
    assert shutil.which("bash") is not None
