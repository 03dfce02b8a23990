"""Developer aid: run golden cases on the GPU stage by stage, each in its own
process with a short timeout, printing every mismatch (no early exit)."""
import os, sys, subprocess, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def stage_align(band):
    from conftest import load_golden
    from falcon_amd.engine import Engine
    eng = Engine(0)
    F3 = load_golden("f3_align")["cases"]
    for c in [c for c in F3 if c["band"] == band]:
        print("  ..", c["name"], len(c["q"]), len(c["t"]), flush=True)
        r, = eng.align_pairs([(c["q"], c["t"])], band=band, want_str=True)
        bad = {k: (r[k], v) for k, v in c["expect"].items() if "str" not in k and r[k] != v}
        sbad = [k for k in ("q_aln_str", "t_aln_str") if c["want_str"] and r[k] != c["expect"][k]]
        print("%-24s %s %s %s" % (c["name"], "OK" if not bad and not sbad else "BAD", bad, sbad), flush=True)

def stage_chain():
    from conftest import load_golden
    from falcon_amd.engine import Engine
    eng = Engine(0)
    F1 = load_golden("f1_f2_hits_ranges")["cases"]
    for c in [c for c in F1 if c["mask"] < 0]:
        print("  ..", c["name"], flush=True)
        b = eng.batch([[c["seed"], c["query"]]]); b.run(4, 8, 0.70)
        r = b.range(1)
        got = [r["s1"], r["e1"], r["s2"], r["e2"], r["score"]]
        print("%-24s %s nhit %d/%d got %s exp %s" % (c["name"], "OK" if got == c["range_48_5"] and r["n_hit"] == c["count"] else "BAD", r["n_hit"], c["count"], got, c["range_48_5"]), flush=True)
        b.free()

def stage_piles():
    from conftest import load_golden
    from helpers import sha_ints
    from falcon_amd.engine import Engine
    eng = Engine(0)
    for c in load_golden("f4_piles")["cases"]:
        print("  ..", c["name"], flush=True)
        try:
            (seq, eqv), = eng.consensus([c["seqs"]], c["min_cov"], 8, c["min_idt"], want_eqv=True)
        except Exception as e:
            print("%-28s EXC %s" % (c["name"], e), flush=True); continue
        ok = seq == c["sequence"] and sha_ints(eqv) == c["eqv_sha"]
        msg = ""
        if not ok:
            n = min(len(seq), len(c["sequence"]))
            first = next((i for i in range(n) if seq[i] != c["sequence"][i]), n)
            msg = "len %d vs %d first diff at %d eqv_ok %s" % (len(seq), len(c["sequence"]), first, sha_ints(eqv) == c["eqv_sha"])
        print("%-28s %s %s" % (c["name"], "OK" if ok else "BAD", msg), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        {"align150": lambda: stage_align(150), "align20": lambda: stage_align(20),
         "chain": stage_chain, "piles": stage_piles}[sys.argv[1]]()
    else:
        for st in ("align150", "align20", "chain", "piles"):
            print("==", st, flush=True)
            try:
                p = subprocess.run([sys.executable, "-u", __file__, st], timeout=int(os.environ.get("STAGE_TIMEOUT", "40")))
                print("== %s rc=%d" % (st, p.returncode), flush=True)
            except subprocess.TimeoutExpired:
                print("== %s TIMEOUT" % st, flush=True)
