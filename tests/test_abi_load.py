"""CPU-side checks of the product library: it builds, loads, and exports every
symbol include/falcon_amd.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "falcon_amd", "libfalcon_amd.so")
HDR = os.path.join(ROOT, "include", "falcon_amd.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(SO)


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-zA-Z_][a-zA-Z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_declares_the_legacy_abi():
    from oracle.pyoracle import LEGACY_SYMBOLS
    names = declared_functions()
    for s in LEGACY_SYMBOLS:
        assert s in names, s


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 19 + 10
    for n in names:
        assert hasattr(lib, n), "libfalcon_amd.so does not export %s" % n


def test_no_gpu_means_loud_failure(lib):
    """Without a HIP device the engine must refuse, not fall back."""
    lib.fa_device_count.restype = ctypes.c_int
    if lib.fa_device_count() > 0:
        pytest.skip("a GPU is visible here")
    lib.fa_create.restype = ctypes.c_void_p
    lib.fa_last_error.restype = ctypes.c_char_p
    assert not lib.fa_create(0)
    assert b"no CPU fallback" in lib.fa_last_error()


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", SO], capture_output=True, text=True).stdout
    assert "oracle" not in out and "falcon_ref" not in out
    syms = subprocess.run(["nm", "-D", SO], capture_output=True, text=True).stdout
    assert " fo_" not in syms


def test_legacy_table_functions_match_golden(lib):
    """The host-struct kup functions of the legacy ABI (not on the GPU hot path)."""
    from conftest import load_golden
    from helpers import check_hits_case
    from oracle.pyoracle import LegacyABI
    impl = LegacyABI(SO)
    for c in load_golden("f1_f2_hits_ranges")["cases"]:
        check_hits_case(impl, c)
    for c in load_golden("f2_ranges_extra")["cases"]:
        assert list(impl.best_range(c["q"], c["t"], c["bin"], c["th"])) == c["range"]


def test_no_asm_block_names_vcc():
    """Inline asm is opaque to the compiler's hazard recognizer.  One hazard it would otherwise
    cover: VCC written implicitly (v_cmp e32) and read by its SGPR number in the next
    instruction needs a wait state ("mixed use of VCC", CDNA3 ISA 4.5) -- which is what an
    asm block with an "s" operand does if the register allocator hands it VCC (seen once:
    two piles of the differential campaign off by a link index).  The kernels keep such
    operands out of asm blocks or behind a compiler-visible instruction; this checks the ISA
    the current sources compile to."""
    import glob
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src_dir = os.path.join(ROOT, "falcon_amd", "csrc")
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for src in sorted(glob.glob(os.path.join(src_dir, "k_*.hip"))):
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            procs.append((src, out, subprocess.Popen(
                [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        for src, out, proc in procs:
            assert proc.wait() == 0, src
            inside = False
            for n, line in enumerate(open(out), 1):
                if "#ASMSTART" in line:
                    inside, by_hand = True, False
                elif "#ASMEND" in line:
                    inside = False
                elif inside and (".La2r_" in line or ".Lwch_" in line or ".Lwc2_" in line):
                    # the hand-scheduled streams (k_align2_rows.h, w_chain, w_chain2), known by their
                    # labels: VCC named in the text and in the clobber list (checked below).
                    by_hand = True
                elif inside and "vcc" in line and not by_hand:
                    # The hazard is a VALU instruction reading VCC as an ordinary SGPR operand
                    # right behind an implicit write.  Two uses are sound and deliberate
                    # (w_row_tail names VCC itself and declares it clobbered, so no operand of
                    # the compiler's choosing can be VCC in that block): a VOPC e32 compare
                    # writing it, and a scalar instruction reading it.
                    text = line.strip()
                    writes = re.match(r"v_cmp_\w+_e32 vcc, (.*)$", text)
                    if writes and "vcc" not in writes.group(1):
                        continue
                    if text.startswith("s_") and not text.startswith("s_nop"):
                        continue
                    bad.append("%s:%d: %s" % (os.path.basename(out), n, text))
    assert not bad, bad
    by_hand = []
    for name in ("k_align2_rows.h", "fa_wave.h"):
        text = open(os.path.join(src_dir, name)).read()
        by_hand += [st for st in re.findall(r"asm volatile\(.*?\);", text, re.S)
                    if ".Lwch_" in st or ".Lwc2_" in st or st.startswith("asm volatile(A2R_BODY")]
    assert len(by_hand) >= 4 and all('"vcc"' in st.rsplit(":", 1)[1] for st in by_hand)


def test_legacy_window_functions_vs_reference_random(lib, ref):
    """find_best_aln_range / find_best_aln_range2 of the product library (host code written in
    this repo's own formulation: sorted diagonals, per-start lower-bound search, compacted
    in-window hits) against the compiled reference on random hit lists -- real k-mer hits of
    noisy sequence pairs, masked and unmasked, and synthetic lists with ties and gaps."""
    import numpy as np
    from falcon_amd.synth import codes_to_str, noisy
    from oracle.pyoracle import LegacyABI
    impl = LegacyABI(SO)
    rng = np.random.default_rng(77)
    n_checked = 0
    for it in range(40):
        n = int(rng.integers(200, 4000))
        g = rng.integers(0, 4, n, dtype=np.uint8)
        if it % 5 == 0:  # low complexity: monster buckets, many ties
            g = np.tile(g[: int(rng.integers(3, 40))], n)[:n]
        e = float(rng.choice([0.0, 0.03, 0.1, 0.2]))
        q, t = codes_to_str(noisy(g, rng, e)), codes_to_str(noisy(g, rng, e))
        for mask in (-1, 16):
            h = ref.find_hits(t, q, mask=mask)
            assert impl.find_hits(t, q, mask=mask) == h
            if not h[0]:
                continue
            assert impl.best_range2(*h) == ref.best_range2(*h), (it, mask)
            assert impl.best_range(*h) == ref.best_range(*h), (it, mask)
            n_checked += 1
    for it in range(200):  # synthetic hit lists (query positions ascending as the table emits them)
        n = int(rng.integers(1, 400))
        qp = np.sort(rng.integers(0, int(rng.integers(50, 5000)), n)).tolist()
        tp = (np.array(qp) + rng.integers(-int(rng.integers(1, 600)), 600, n)).clip(0).tolist()
        assert impl.best_range2(qp, tp) == ref.best_range2(qp, tp), it
        n_checked += 1
    assert n_checked > 200
