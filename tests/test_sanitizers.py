"""The product's concurrent HOST-SIDE code under ThreadSanitizer and AddressSanitizer + UBSan, in the GPU-less container.

csrc/gfdm_hostpipe.hip (copy pool, chunk planner, registration registry, completion tickets, staging sets), csrc/gfdm_jit.hip (disk cache, background builds,
quiesce, the exit path), csrc/gfdm_hip_api.hip (handles, family pin, every *_host entry point), the C++ classes, sharded_batch.h (a thread per device) and
batched_work.h are compiled UNCHANGED as plain C++ against the test-only loop-back HIP layer of tests/sanitize/loopback (streams = worker threads, a launch = a
host function that copies in -> out) and driven by tests/sanitize/host_fuzz.cc: the call mix of scratch/fuzz_host_path.py on several threads at once, quiesce
while calls are in flight, every runtime call failing once, the process ending with builds in flight.  Recipe: tests/sanitize/Makefile.

Findings of the first runs (fixed in the product): a stale sticky hipGetLastError() of an earlier failed call failing the next launch, hipEvents leaked when
their creation failed half way, a device operand with an unmapped tail bounced on the CPU instead of refused.
GPU-side sanitizers are not available on this pool; the device code is covered by the parity tests.
"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = os.path.join(ROOT, "tests", "sanitize")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="the ROCm clang (host compiler with the sanitizer runtimes) is not installed")


def build(flavour, out, pkg=None):
    cmd = ["make", "-C", SAN, "-j4", flavour, "OUT=" + str(out)]
    if pkg:
        cmd.append("PKG=" + str(pkg))
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    return os.path.join(str(out), "host_fuzz_" + flavour)


def run(exe, seconds, seed, env_extra, tmp_path):
    env = dict(os.environ, HOME=str(tmp_path), TMPDIR=str(tmp_path), **env_extra)      # the driver's code-object cache goes under TMPDIR
    p = subprocess.run([exe, str(seconds), str(seed), "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600, cwd=str(tmp_path))
    return p.returncode, p.stdout.decode(errors="replace")


TSAN_ENV = {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}
ASAN_ENV = {"ASAN_OPTIONS": "detect_leaks=1 detect_stack_use_after_return=1", "UBSAN_OPTIONS": "print_stacktrace=1 halt_on_error=1"}


def test_host_side_is_clean_under_thread_sanitizer(tmp_path):
    exe = build("tsan", tmp_path / "build")
    rc, out = run(exe, 6, 11, TSAN_ENV, tmp_path)
    assert rc == 0 and "host fuzz OK" in out and "ThreadSanitizer" not in out, out[-6000:]
    assert "edge cases:" in out and "failure sweep:" in out


def test_host_side_is_clean_under_address_and_ub_sanitizer(tmp_path):
    exe = build("asan", tmp_path / "build")
    rc, out = run(exe, 6, 12, ASAN_ENV, tmp_path)
    assert rc == 0 and "host fuzz OK" in out and "Sanitizer" not in out and "runtime error" not in out, out[-6000:]


def mutated_package(tmp_path, path, old, new):
    """a copy of the product sources with one planted bug"""
    pkg = tmp_path / "mut" / "gr-gfdm_amd"
    shutil.copytree(os.path.join(ROOT, "gr-gfdm_amd", "csrc"), pkg / "csrc")
    shutil.copytree(os.path.join(ROOT, "gr-gfdm_amd", "cpp"), pkg / "cpp")
    shutil.copy(os.path.join(ROOT, "gr-gfdm_amd", "Makefile"), pkg / "Makefile")
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "mut" / "include")
    f = pkg / path
    text = f.read_text()
    assert text.count(old) == 1, "the planted-bug anchor moved: %r" % old
    f.write_text(text.replace(old, new))
    return pkg


def test_the_harness_sees_a_planted_race_and_a_planted_overrun(tmp_path):
    """sensitivity: a lock taken out of the registration registry is a ThreadSanitizer report, a staging set eight bytes short an AddressSanitizer report"""
    pkg = mutated_package(tmp_path, "csrc/gfdm_hostpipe.hip", "    std::lock_guard<std::mutex> lk(g_reg_mu);\n    for (const auto& r : g_registered)",
                          "    for (const auto& r : g_registered)")
    exe = build("tsan", tmp_path / "b1", pkg)
    for seed in (1, 2, 3):                          # (a race is caught when two threads really meet on the registry: every run so far did, a second seed is insurance)
        rc, out = run(exe, 4, seed, TSAN_ENV, tmp_path)
        if rc != 0 and "ThreadSanitizer: data race" in out and "registry_contains" in out:
            break
    else:
        pytest.fail("the planted race was not reported:\n" + out[-3000:])
    shutil.rmtree(tmp_path / "mut")
    pkg = mutated_package(tmp_path, "csrc/gfdm_hostpipe.hip", "const size_t sz = extent[i] ? align_up(chunk_size(i, chunk_blocks)) : 0;",
                          "const size_t sz = extent[i] ? chunk_size(i, chunk_blocks) - 8 : 0;")
    rc, out = run(build("asan", tmp_path / "b2", pkg), 4, 1, ASAN_ENV, tmp_path)
    assert rc != 0 and "AddressSanitizer: heap-buffer-overflow" in out, out[-3000:]
