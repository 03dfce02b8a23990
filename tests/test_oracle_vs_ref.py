"""Randomised differential test: our restatement vs the compiled reference C
(oracle/_ref).  Skipped where the reference build is absent."""
import numpy as np

from falcon_amd.synth import codes_to_str, make_pile, noisy, pile_to_seqs


def test_functions_random(port, ref):
    rng = np.random.default_rng(2024)
    for _ in range(25):
        n = int(rng.integers(30, 3000))
        g = rng.integers(0, 4, n, dtype=np.uint8)
        e = float(rng.choice([0.0, 0.05, 0.13, 0.25, 0.4]))
        q, t = codes_to_str(noisy(g, rng, e)), codes_to_str(noisy(g, rng, e))
        if len(q) < 10 or len(t) < 10:
            continue
        hp = port.find_hits(t, q)
        assert hp == ref.find_hits(t, q)
        assert port.best_range(*hp) == ref.best_range(*hp)
        hm = port.find_hits(t, q, mask=16)
        assert hm == ref.find_hits(t, q, mask=16)
        if hm[0]:
            assert port.best_range2(*hm) == ref.best_range2(*hm)
        for band in (150, 20, 1500):
            a = port.align(q, t, band, 1)
            a.pop("cells")
            assert a == ref.align(q, t, band, 1)


def test_piles_random(port, ref):
    for seed, (S, cov, het) in enumerate([(2000, 15, 0.0), (5000, 15, 0.0), (5000, 20, 0.01)]):
        s, rd = make_pile(seed + 3, S=S, coverage=cov, het=het, min_read=500,
                          mean_read=S * 0.6, sd_read=S * 0.2)
        seqs = [codes_to_str(x) for x in pile_to_seqs(s, rd)]
        for min_cov, idt in ((4, 0.70), (0, 0.80)):
            assert port.generate_consensus(seqs, min_cov, 8, idt) == \
                ref.generate_consensus(seqs, min_cov, 8, idt)


def _pile_with_long_insertions(run_lengths, seed=77):
    """A clean pile whose reads carry long inserted blocks (deep MSA insertion levels)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, 3000, dtype=np.uint8)
    seed_read = noisy(g, rng, 0.02)
    reads = []
    for i in range(14):
        r = noisy(g, rng, 0.03)
        if i < len(run_lengths) * 2:
            n = run_lengths[i % len(run_lengths)]
            at = 900 + 150 * (i % len(run_lengths))
            r = np.concatenate([r[:at], rng.integers(0, 4, n, dtype=np.uint8), r[at:]])
        reads.append(r)
    return [codes_to_str(x) for x in pile_to_seqs(seed_read, reads)]


def test_long_insertion_runs(port, ref):
    """Insertion runs beyond the inline capacity of a GPU tag (16) but inside the
    reference's parity domain (< 248)."""
    pile = _pile_with_long_insertions([20, 40, 120, 200])
    assert port.generate_consensus(pile, 2, 8, 0.70) == ref.generate_consensus(pile, 2, 8, 0.70)
