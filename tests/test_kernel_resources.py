"""What the kernels' launch shapes rely on, read off the compiler's output (hipcc -S for gfx950, no GPU):
LDS bytes and VGPRs per wavefront decide how many wavefronts a SIMD holds, and several kernels were sized
for that on purpose -- a table that grows by a few bytes would silently cost a
wavefront per SIMD.  160 KB of LDS per CU, 512 VGPRs per SIMD lane, at most 8 wavefronts per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LDS_PER_CU = 160 * 1024

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")


def resources(src):
    """{kernel name: (LDS bytes, VGPRs)} of a .hip file."""
    asm = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-",
                          os.path.join(CSRC, src)], capture_output=True, text=True, check=True, cwd=CSRC).stdout
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        v = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(name), asm)
        out[name] = (lds, int(v.group(1)) if v else None)
    return out


def waves_per_simd(lds, vgpr, waves_per_group=1):
    by_lds = (LDS_PER_CU // max(lds, 1)) * waves_per_group // 4
    by_vgpr = 512 // (-(-vgpr // 8) * 8)
    return min(8, by_lds, by_vgpr)


def test_k_links2_holds_eight_wavefronts_per_simd():
    r = resources("k_links2.hip")
    (name, (lds, vgpr)), = [(k, v) for k, v in r.items() if "k_links2" in k and "big" not in k]
    assert lds <= 5120 and vgpr <= 64, (name, lds, vgpr)   # 10.65 -> 9.34 ms came from exactly this
    assert waves_per_simd(lds, vgpr) == 8


def test_per_pile_kernels_hold_a_bench_batch_at_once():
    """k_score2 (two wavefronts per pile) and k_backtrace run a pile per workgroup from start to end: the
    3072 piles of a bench batch are 12 workgroups per CU, which must all be resident."""
    (lds, vgpr), = [v for k, v in resources("k_score2.hip").items() if "k_score2" in k]
    assert LDS_PER_CU // lds >= 12 and 12 * 2 <= 4 * (512 // (-(-vgpr // 8) * 8)), (lds, vgpr)
    (lds, vgpr), = [v for k, v in resources("k_msa.hip").items() if "k_backtrace" in k]
    assert LDS_PER_CU // lds >= 12 and 12 <= 4 * (512 // (-(-vgpr // 8) * 8)), (lds, vgpr)
    (lds, vgpr), = [v for k, v in resources("k_msa.hip").items() if "k_tags" in k]
    assert waves_per_simd(lds, vgpr) == 8, (lds, vgpr)


def test_the_seed_index_table_fits_one_cu():
    r = resources("k_seed_index.hip")
    (lds, vgpr), = [v for k, v in r.items() if "k_seed_index" in k and "long" not in k]
    assert 128 * 1024 < lds <= LDS_PER_CU and 4 * (-(-vgpr // 8) * 8) <= 512, (lds, vgpr)   # 16 wavefronts of one workgroup


def test_the_chaining_kernel_keeps_eight_wavefronts_per_simd():
    """k_chain hides a string of dependent reads behind its occupancy: more than 64 vector registers would cost it a
    wavefront per SIMD (512 registers per lane and SIMD).  Round 6's first pass, with its five packed-read loads issued
    together, needs 60."""
    r = resources("k_chain.hip")
    (lds, vgpr), = [v for k, v in r.items() if "k_chain" in k]
    assert vgpr <= 64, vgpr


def _listing(src, *flags):
    return subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-",
                           *flags, os.path.join(CSRC, src)], capture_output=True, text=True, check=True, cwd=CSRC).stdout


def test_what_the_unmeasured_build_switches_change_in_the_listings():
    """Round 6 left two build switches OFF because the GPU was closed before they could run.
    What each is there for can be read off the listing without one:
    W_ADDR32 -- k_align2's event loop keeps a register tuple for 64-bit array indices in scratch (99 of ~200 scratch
    instructions, 16 bytes per lane each: the 16 GB of writes that were nobody's data); with 32-bit offsets the
    tuple and most 64-bit address additions are gone.
    S2_SETTLE_NL -- k_score2's chain loop waits for the LDS twice per level; with the link counts consumed before
    the loop the wait at the loop's head (for the atomic of the level before) is gone."""
    base, addr32 = _listing("k_align2.hip"), _listing("k_align2.hip", "-DW_ADDR32")
    n = lambda text, pat: len(re.findall(pat, text))
    assert n(base, r"\bscratch_(load|store)") >= 190 and n(addr32, r"\bscratch_(load|store)") <= 140
    assert n(base, r"v_lshl_add_u64") >= 40 and n(addr32, r"v_lshl_add_u64") <= 6

    def chain_loop(text):   # the loop around the chain's ds_max_u32, from its header to its back edge
        lines = text.split("\n")
        at = next(i for i, ln in enumerate(lines) if "ds_max_u32" in ln)
        head = max(i for i in range(at) if lines[i].startswith(".LBB"))
        tail = next(i for i in range(at, len(lines)) if "s_cbranch" in lines[i])
        return lines[head:tail + 1]
    waits = lambda body: sum(1 for ln in body if "s_waitcnt lgkmcnt(0)" in ln)
    assert waits(chain_loop(_listing("k_score2.hip"))) == 2
    assert waits(chain_loop(_listing("k_score2.hip", "-DS2_SETTLE_NL"))) == 1


def test_the_first_pass_of_k_chain_issues_its_packed_read_loads_together():
    """Round 6: written `(p < n_probe) ? fa_kmer8(..) : 0`, each of the five loads of a step sat in a block of its own
    that ended in `s_waitcnt vmcnt(0)` -- five latencies in a row.  With the index clamped instead they are issued
    back to back: in the listing of the first pass's loop, five loads stand before the first wait."""
    lines = _listing("k_chain.hip").split("\n")
    head = next(i for i, ln in enumerate(lines) if "Inner Loop Header: Depth=1" in ln)
    loads = 0
    for ln in lines[head:head + 200]:
        if "global_load" in ln:
            loads += 1
        if "s_waitcnt vmcnt" in ln:
            break
    assert loads >= 5, loads
