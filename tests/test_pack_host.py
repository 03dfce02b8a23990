"""The host-side 2-bit packer (falcon_amd/csrc/pack_host.cpp, eight bases per 64-bit step) against
a plain numpy restatement of the layout k_pack defines: 16 bases per u32, base i at bits
2 (i mod 16), A C G T = 0 1 2 3, zero words behind the sequence; and its verdict on bytes
outside upper-case ACGT (the position of the first one), for every byte value."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from falcon_amd.lib import load
    return load()


def _pack(lib, s: bytes, extra=3):
    n_out = (len(s) + 15) // 16 + extra
    out = (C.c_uint * max(n_out, 1))(*([0xDEADBEEF] * max(n_out, 1)))
    rc = lib.fa_debug_pack(s, len(s), out, n_out)
    return rc, np.array(out[:n_out], dtype=np.uint32)


def _want(s: bytes, extra=3):
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    v = code[np.frombuffer(s, dtype=np.uint8)].astype(np.uint64)
    nw = (len(s) + 15) // 16
    pad = np.zeros(nw * 16, dtype=np.uint64)
    pad[:len(s)] = v
    words = (pad.reshape(nw, 16) << (2 * np.arange(16, dtype=np.uint64))[None, :]).sum(axis=1) if nw else np.zeros(0)
    return np.concatenate([words.astype(np.uint32), np.zeros(extra, dtype=np.uint32)])


def test_packed_words_of_random_sequences(lib):
    rng = np.random.default_rng(5)
    for n in list(range(0, 70)) + [127, 128, 129, 1000, 4097, 65536 + 5]:
        s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
        rc, got = _pack(lib, s)
        assert rc == -1, n
        assert (got == _want(s)).all(), n


def test_every_byte_value_at_every_phase(lib):
    """A byte is accepted iff it is one of A C G T; the position reported is the FIRST such byte
    -- in each of the 16 positions of a word, in the 8-byte halves and in the scalar tail."""
    base = b"ACGTTGCAGTCAACGT" * 3 + b"ACGTA"   # 53 bases: three full words + a tail of 5
    for c in range(256):
        for at in (0, 7, 8, 15, 16, 31, 47, 48, 52):
            s = bytearray(base)
            s[at] = c
            rc, _ = _pack(lib, bytes(s))
            assert rc == (-1 if bytes([c]) in (b"A", b"C", b"G", b"T") else at), (c, at)
    s = bytearray(base)
    s[20], s[40], s[50] = ord("N"), ord("a"), 0
    assert _pack(lib, bytes(s))[0] == 20


def test_bad_arguments(lib):
    out = (C.c_uint * 1)()
    assert lib.fa_debug_pack(b"ACGT" * 8, 32, out, 1) == -2  # (32 bases need two words)
