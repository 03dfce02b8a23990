"""Shared golden-vector checkers: the same assertions are applied to the CPU
oracle (not gpu) and to the HIP product library through its C ABI (gpu)."""
import hashlib

import numpy as np


def sha_ints(xs):
    return hashlib.sha1(np.asarray(xs, dtype="<i4").tobytes()).hexdigest()


def check_hits_case(impl, c):
    q, t = impl.find_hits(c["seed"], c["query"], c["K"], c["mask"])
    assert len(q) == c["count"], c["name"]
    assert sha_ints(q) == c["sha_q"] and sha_ints(t) == c["sha_t"], c["name"]
    if "q" in c:
        assert q == c["q"] and t == c["t"], c["name"]
    assert list(impl.best_range(q, t, 48, 5)) == c["range_48_5"], c["name"]
    assert list(impl.best_range(q, t, 80, 50)) == c["range_80_50"], c["name"]
    if "range2" in c:
        assert list(impl.best_range2(q, t)) == c["range2"], c["name"]


def check_align_case(impl, c):
    a = impl.align(c["q"], c["t"], c["band"], c["want_str"])
    a.pop("cells", None)
    for k, v in c["expect"].items():
        assert a[k] == v, (c["name"], k)


def check_pile_case(impl, c):
    seq, eqv = impl.generate_consensus(c["seqs"], c["min_cov"], c["K"], c["min_idt"])
    assert seq == c["sequence"], c["name"]
    assert eqv[:64] == c["eqv_head"], c["name"]
    assert sha_ints(eqv) == c["eqv_sha"], c["name"]


def config_pile(c):
    """The full-size pile of a benchmark configuration (tests/golden/f8_configs): the
    input is regenerated from the generator parameters and checked against its digest."""
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    seed, reads = make_pile(c["seed"], S=c["S"], coverage=c["coverage"], het=c["het"])
    seqs = [codes_to_str(x) for x in pile_to_seqs(seed, reads, 200)]
    assert len(seqs) == c["n_seq"] and sum(map(len, seqs)) == c["n_bases"], c["name"]
    assert hashlib.sha1("\n".join(seqs).encode()).hexdigest() == c["input_sha"], c["name"]
    return seqs


def check_config_case(impl, c):
    seq, eqv = impl.generate_consensus(config_pile(c), c["min_cov"], c["K"], c["min_idt"])
    assert seq == c["sequence"], c["name"]
    assert sha_ints(eqv) == c["eqv_sha"], c["name"]
