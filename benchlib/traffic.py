"""bench.py, what the roofline object says about HBM: the rate a plain device copy reaches on this box, and the dominant
kernel's measured HBM bytes per launch -- from the rocprofv3 PMC passes committed under profiles/, and only when they were
taken on the kernel's CURRENT source."""
from __future__ import annotations

import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measured_stream_rate(torch, mib=1024, reps=5):
    """GB/s (read + write) of a device-to-device copy of `mib` MiB: what HBM delivers to the
    simplest streaming kernel on this box (outside the timed region; plumbing, not product)."""
    try:
        a = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del a, b
        return round(2.0 * (mib << 20) / (ms * 1e-3) / 1e9, 1)
    except Exception:  # informative only
        return None


KERNEL_SOURCE = {"k_align": ("k_align2.hip", "k_align2_core.h", "k_align2_rows.h", "fa_wave.h", "k_align.hip"),
                 "k_score": ("k_score2.hip", "k_score1.hip", "k_msa.h"), "k_links": ("k_links2.hip", "k_msa.hip", "k_msa.h"),
                 "k_tags": ("k_msa.hip", "k_msa.h"), "k_backtrace": ("k_msa.hip", "k_msa.h"), "k_chain": "k_chain.hip",
                 "k_seed_index": "k_seed_index.hip"}


def _code_only(text):
    """C++ source without comments and blank space: what the digest below is taken of (a
    reworded comment does not make a measurement stale; string literals -- the inline asm --
    are kept as they are)."""
    import re
    pat = re.compile(r'"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)
    text = pat.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return "\n".join(" ".join(ln.split()) for ln in text.splitlines() if ln.strip())


def kernel_source_sha(kernel):
    """Digest of the code (comments and layout aside) of the source files a kernel lives in:
    a PMC measurement is only presented as this build's when it was taken on this code."""
    try:
        names = KERNEL_SOURCE[kernel]
        h = hashlib.sha1()
        for name in ((names,) if isinstance(names, str) else names):
            with open(os.path.join(ROOT, "falcon_amd", "csrc", name), "r", errors="replace") as f:
                h.update(_code_only(f.read()).encode())
        return h.hexdigest()[:16]
    except (KeyError, OSError):
        return None


def measured_traffic(kernel, piles, workload):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under
    profiles/ (the TCC counters cannot be read from inside this process); None unless a
    measurement of this kernel's CURRENT source, at this batch size and workload, is on file
    (profiles/pmc_traffic.json, written by scripts/pmc_traffic_record.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        rec = doc.get("%s:%s" % (kernel, workload)) or doc.get(kernel)
        if not rec:
            return None, "no PMC measurement of this kernel under profiles/"
        if int(rec["piles_per_launch"]) != int(piles) or rec.get("workload", "ecoli") != workload:
            return None, "the PMC measurement on file is of another batch size or workload"
        if rec.get("source_sha") != kernel_source_sha(kernel):
            return None, ("the PMC measurement on file (%s) was taken on an older source of this "
                          "kernel" % rec.get("taken_on", "?"))
        return int(rec["hbm_bytes_per_launch"]), rec.get("what", "")
    except Exception as e:
        return None, "profiles/pmc_traffic.json unreadable: %r" % (e,)


# what the stages' kernels are called in a rocprofv3 trace (bench.py's keys are the path's stages; the
# kernels behind them -- FALCON_AMD_ALIGN1 / _LINKS1 / _SCORE1 -- keep the stage's name)
KERNEL_NAME = {"k_align": "k_align2", "k_links": "k_links2", "k_score": "k_score2"}

# what a CU of this chip issues at most, vector + scalar instructions of co-resident wavefronts added up
# (profiles/r02_ubench_issue_rates.txt, scripts/ubench/issue_rates.hip: 1058 G/s vector alone, 610 G/s
# scalar alone, 1192 G/s of both together)
ISSUE_CEILING_G_PER_S = {"valu": 1058.0, "salu": 610.0, "valu+salu": 1192.0}


def measured_issue(kernel, piles, workload, launch_ms):
    """The instruction-issue side of the dominant kernel, from the SQ_INSTS_* passes on file
    (profiles/pmc_issue.json, scripts/pmc_issue_record.py): wave-instructions per launch by class,
    and -- with THIS run's launch time -- the rate against the measured ceiling.  None unless the
    passes were taken on the kernel's current source at this batch size and workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_issue.json")) as f:
            rec = json.load(f).get(kernel)
        if not rec:
            return None
        if int(rec["piles_per_launch"]) != int(piles) or rec.get("workload", "ecoli") != workload:
            return None
        if rec.get("source_sha") != kernel_source_sha(kernel):
            return None
        v, sc = float(rec["valu"]), float(rec["salu"])
        rate = (v + sc) / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        return {"valu": int(v), "salu": int(sc), "lds": None if rec.get("lds") is None else int(rec["lds"]), "vmem": int(rec.get("vmem", 0)),
                "unit": "wave-instructions per launch",
                "wave_instr_per_s_G": round(rate, 1),
                "ceiling_measured_G": ISSUE_CEILING_G_PER_S["valu+salu"],
                "frac": round(rate / ISSUE_CEILING_G_PER_S["valu+salu"], 4),
                "valu_frac": round(v / (launch_ms * 1e-3) / 1e9 / ISSUE_CEILING_G_PER_S["valu"], 4),
                "salu_frac": round(sc / (launch_ms * 1e-3) / 1e9 / ISSUE_CEILING_G_PER_S["salu"], 4),
                "source": rec.get("what", ""), "taken_on": rec.get("taken_on")}
    except Exception:
        return None
