"""The HIP-free host side of the library -- the stream reader with its read helpers and its ring of kept
batches (csrc/reader.cpp), host packing from several threads (csrc/pack_host.cpp), the record printer
(csrc/fasta.cpp) -- under ThreadSanitizer and AddressSanitizer + UBSan (tests/san/host_san.cpp drives them the
way the worker does: an ingest thread running ahead of a staging thread, file and ragged pipe, 2 and 8 batches
kept, one and four read helpers).  Any report ends the binary with a status other than 0."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon_amd", "csrc")


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_host_side_under_a_sanitizer(kind):
    b = subprocess.run(["make", "-s", "-C", CSRC, kind], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode(errors="replace")[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([os.path.join(ROOT, "tests", "san", "host_" + kind), "48"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "no difference" in out, out[-4000:]
    assert "Sanitizer" not in out, out[-4000:]


def test_the_batch_engine_s_host_logic_under_threadsanitizer():
    """csrc/engine.hip is pure host code -- staging, the front half under the context's lock, the planner thread,
    the back half, fetch, results, freeing, the block cache.  ThreadSanitizer cannot load the HIP runtime, so the
    engine is compiled against stand-ins for the runtime and for the kernel launchers (tests/san/stub,
    engine_stub_kernels.cpp: no alignment is ever accepted) and driven like the worker drives it: three runner
    threads on one context, two on another, up to three batches per thread between submit and wait, batches
    freed while in flight, batches run again (tests/san/engine_san.cpp)."""
    b = subprocess.run(["make", "-s", "-C", CSRC, "tsan_engine"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode(errors="replace")[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    p = subprocess.run([os.path.join(ROOT, "tests", "san", "engine_tsan"), "80"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "nothing failed" in out, out[-4000:]
    assert "Sanitizer" not in out, out[-4000:]
