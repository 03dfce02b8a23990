"""The native LA4Falcon stream reader (falcon_amd/csrc/reader.cpp) against
the python restatement of get_seq_data / get_longest_reads
(/root/reference/falcon_kit/mains/consensus.py:161-209, :26-45) that the CLI tests pin
against the reference's own driver (tests/golden/f5_cli).  No GPU needed."""
import io
import os
import random
import tempfile

import pytest

from falcon_amd.engine import Reader
from falcon_amd.mains.consensus import PileReader, Settings, _run_native, parse_args, settings_from


def _native(text, min_n_read, min_len_aln, min_cov_aln, max_n_read, max_cov_aln, max_piles=0,
            max_bases=0, raw=False):
    with tempfile.NamedTemporaryFile("wb", delete=False) as f:
        f.write(text if raw else text.encode("ascii"))
        path = f.name
    fd = os.open(path, os.O_RDONLY)
    try:
        r = Reader(fd, min_n_read, min_len_aln, min_cov_aln, max_n_read, max_cov_aln)
        out, calls = [], 0
        while True:
            ps = r.next(max_piles, max_bases)
            if ps is None:
                break
            calls += 1
            out.extend(zip(ps.seed_ids, ps.piles()))
        r.close()
        return out, calls
    finally:
        os.close(fd)
        os.unlink(path)


def _python(text, min_n_read, min_len_aln, min_cov_aln, max_n_read, max_cov_aln):
    cfg = Settings(4, 8, max_n_read, 0.70, 1000, 50, min_cov_aln, max_cov_aln)
    # newline=None: universal newlines, like the text-mode sys.stdin the worker iterates
    return list(PileReader(io.StringIO(text, newline=None), cfg, min_n_read, min_len_aln))


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _rand_stream(rng, n_pile, with_noise):
    lines = []
    for p in range(n_pile):
        n_read = rng.randint(1, 14)
        seed_len = rng.randint(30, 200)
        names = ["%06d" % rng.randint(0, 40) for _ in range(n_read)]
        for i, nm in enumerate(names):
            ln = seed_len if i == 0 else rng.randint(5, 260)
            lines.append("%s %s" % (nm, _rand_seq(rng, ln)))
            if with_noise and rng.random() < 0.08:
                lines.append(rng.choice(["", "   ", "justonetoken", "a b c", "\t", "x  y  z w"]))
            if with_noise and rng.random() < 0.05:
                lines.append("%s\t %s  " % ("%06d" % rng.randint(0, 40), _rand_seq(rng, rng.randint(5, 90))))
        lines.append(rng.choice(["+ +", "+ +", "+ +", "* *", "+ anything"]))
    lines.append("- -")
    lines.append("999999 ACGTACGT")  # after the end marker: never read
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", range(12))
def test_reader_equals_python_parser(seed):
    rng = random.Random(seed)
    text = _rand_stream(rng, rng.randint(1, 25), with_noise=seed % 2 == 1)
    for opts in [(1, 0, 0, 500, 0), (4, 0, 1, 500, 0), (3, 40, 1, 6, 0), (2, 0, 0, 5, 2),
                 (10, 0, 10, 200, 0)]:
        want = _python(text, *opts)
        got, _ = _native(text, *opts)
        assert got == want, opts


def test_batch_limits_do_not_change_the_piles():
    rng = random.Random(77)
    text = _rand_stream(rng, 40, with_noise=True)
    opts = (2, 0, 0, 500, 0)
    want = _python(text, *opts)
    one, c1 = _native(text, *opts)
    by3, c3 = _native(text, *opts, max_piles=3)
    byb, cb = _native(text, *opts, max_bases=700)
    assert one == want and by3 == want and byb == want
    assert c1 == 1 and c3 == -(-len(want) // 3) and cb > 1


def test_a_batch_outlives_one_further_call():
    """include/falcon_amd.h: what fa_reader_next hands out stays valid through one further
    call (the worker stages batch n while it reads batch n + 1)."""
    rng = random.Random(78)
    text = _rand_stream(rng, 60, with_noise=True)
    opts = (2, 0, 0, 500, 0)
    want = _python(text, *opts)
    with tempfile.NamedTemporaryFile("wb", delete=False) as f:
        f.write(text.encode("ascii"))
        path = f.name
    fd = os.open(path, os.O_RDONLY)
    try:
        for limits in ((1, 0), (4, 0), (0, 900)):
            os.lseek(fd, 0, os.SEEK_SET)
            r = Reader(fd, *opts)
            got, prev = [], None
            while True:
                ps = r.next(*limits)
                if prev is not None:  # read only now, after the call that followed it
                    got.extend(zip(prev.seed_ids, prev.piles()))
                    assert [prev.seed_ids[i] for i in range(prev.n_pile)] == \
                           [prev.raw_ids[i].decode("ascii") for i in range(prev.n_pile)]
                if ps is None:
                    break
                prev = ps
            r.close()
            assert got == want
    finally:
        os.close(fd)
        os.unlink(path)


@pytest.mark.parametrize("keep", [3, 8])
def test_batches_outlive_as_many_calls_as_asked_for(keep):
    """fa_reader_keep(n): a batch lives through the next n - 1 calls (the worker's ingest thread runs
    that far ahead of the staging thread); every batch is read only when it is about to lapse."""
    rng = random.Random(79)
    text = _rand_stream(rng, 90, with_noise=True)
    opts = (2, 0, 0, 500, 0)
    want = _python(text, *opts)
    with tempfile.NamedTemporaryFile("wb", delete=False) as f:
        f.write(text.encode("ascii"))
        path = f.name
    fd = os.open(path, os.O_RDONLY)
    try:
        for limits in ((1, 0), (3, 0), (0, 900)):
            os.lseek(fd, 0, os.SEEK_SET)
            r = Reader(fd, *opts)
            r.keep(keep)
            got, held = [], []
            while True:
                ps = r.next(*limits)
                if ps is not None:
                    held.append(ps)
                while held and (ps is None or len(held) == keep):
                    old = held.pop(0)  # handed out keep - 1 calls ago
                    got.extend(zip(old.seed_ids, old.piles()))
                if ps is None:
                    break
            with pytest.raises(Exception):
                r.keep(2)  # only before the first batch
            r.close()
            assert got == want
    finally:
        os.close(fd)
        os.unlink(path)


def test_closing_the_reader_does_not_wait_for_a_silent_producer():
    """The reader keeps one read() ahead of the scanner on a helper thread (reader.cpp).  A
    producer that has delivered a batch and then says nothing (the pipe stays open) must not
    hold up that batch, nor the close that follows while the read-ahead is still waiting."""
    import time
    rng = random.Random(6)
    text = _rand_stream(rng, 30, with_noise=False).split("- -")[0]  # (no end marker, no EOF)
    opts = (2, 0, 0, 500, 0)
    want = _python(text + "- -\n", *opts)
    rd, wr = os.pipe()
    try:
        os.write(wr, text.encode("ascii"))
        r = Reader(rd, *opts)
        t0 = time.perf_counter()
        ps = r.next(len(want), 0)  # exactly what has been written: no need to wait for more
        got = list(zip(ps.seed_ids, ps.piles()))
        r.close()
        assert time.perf_counter() - t0 < 2.0
        assert got == want
    finally:
        os.close(wr)
        os.close(rd)


def test_pipe_with_ragged_writes_and_control_bytes():
    """The scanner keeps its per-line state across short reads (a pipe fed in odd-sized
    pieces), across 16-byte compare blocks and across batch boundaries; control bytes that
    are not white space do not split tokens."""
    import threading
    rng = random.Random(5)
    text = _rand_stream(rng, 60, with_noise=True)
    # bytes <= 0x20 that str.split() does not treat as separators, inside a name / alone
    text = "na\x01me ACGTACGTAC\nr\x00x ACGTACGTACGT\n\x02\n+ +\n" + text
    data = text.encode("latin-1")
    opts = (2, 0, 0, 500, 0)
    want = _python(text, *opts)
    for max_bases in (0, 300):
        rd, wr = os.pipe()

        def feed():
            i = 0
            try:
                while i < len(data):
                    n = rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 64, 257, 4099])
                    os.write(wr, data[i:i + n])
                    i += n
            except BrokenPipeError:  # the reader stops at "- -" and may close first
                pass
            os.close(wr)

        t = threading.Thread(target=feed)
        t.start()
        r = Reader(rd, *opts)
        got = []
        while True:
            ps = r.next(0, max_bases)
            if ps is None:
                break
            got.extend(zip(ps.seed_ids, ps.piles()))
        r.close()
        os.close(rd)   # (a writer still blocked after "- -" gets EPIPE and ends)
        t.join()
        assert got == want


def test_a_pipe_with_eight_batches_kept():
    """The worker keeps 8 batches alive (fa_reader_keep) whatever the stream is -- a pipe from
    LA4Falcon in production: ragged writes, small batches, every batch read only when it is about to
    lapse."""
    import threading
    rng = random.Random(6)
    text = _rand_stream(rng, 80, with_noise=True)
    data = text.encode("latin-1")
    opts = (2, 0, 0, 500, 0)
    want = _python(text, *opts)
    rd, wr = os.pipe()

    def feed():
        i = 0
        try:
            while i < len(data):
                n = rng.choice([1, 5, 16, 63, 64, 65, 300, 5000])
                os.write(wr, data[i:i + n])
                i += n
        except BrokenPipeError:
            pass
        os.close(wr)

    t = threading.Thread(target=feed)
    t.start()
    r = Reader(rd, *opts)
    r.keep(8)
    got, held = [], []
    while True:
        ps = r.next(2, 0)
        if ps is not None:
            held.append(ps)
        while held and (ps is None or len(held) == 8):
            old = held.pop(0)
            got.extend(zip(old.seed_ids, old.piles()))
        if ps is None:
            break
    r.close()
    os.close(rd)
    t.join()
    assert got == want


def test_long_sequences_are_cut_and_stream_end_variants():
    big = "A" * 100001
    edge = "C" * 100000
    text = "s1 %s\nr1 %s\nr2 ACGT\n+ +\n" % (big, edge)
    for tail in ["- -\n", "", "s2 ACGT"]:  # explicit end, bare EOF, EOF inside a pile
        want = _python(text + tail, 1, 0, 0, 500, 0)
        got, _ = _native(text + tail, 1, 0, 0, 500, 0)
        assert got == want
        assert len(got) == 1 and len(got[0][1][0]) == 99999 and len(got[0][1][1]) == 100000
    # CRLF line ends and a last line without newline
    crlf = "s1 ACGTACGT\r\nr1 ACGTAC\r\n+ +\r\ns2 AAAA\r\nr9 CC\r\n+ +"
    assert _native(crlf, 1, 0, 0, 500, 0)[0] == _python(crlf, 1, 0, 0, 500, 0)


def test_lone_carriage_returns_end_lines_like_universal_newlines():
    """A text-mode sys.stdin (what consensus.py:170 iterates) ends lines at '\\n', '\\r\\n' and a
    lone '\\r'; the native reader agrees byte for byte, also when the '\\r' is the last byte
    of a read() or of the stream."""
    opts = (2, 0, 0, 500, 0)
    for text in ("s1 ACGTACGT\rr1 ACGTAC\r+ +\rs2 AAAA\rr9 CC\r+ +\r",
                 "s1 ACGTACGT\rr1 ACGTAC\r\n+ +\r\r\ns2 AAAA\n\rr9 CC\r+ +",
                 "s1 ACGT\r", "\r", "s1 AC\rr1 GG\r+ +\r- -\rs2 AAAA\rr9 CC\r+ +\r"):
        got, _ = _native(text, *opts)
        assert got == _python(text, *opts), repr(text)
    # the '\r' falls on the end of a pipe read: fed byte by byte
    text = "s1 ACGTACGT\rr1 ACGTAC\r+ +\rs2 AAAA\r\nr9 CC\r+ +\r"
    rd, wr = os.pipe()

    def feed():
        for ch in text.encode():
            os.write(wr, bytes([ch]))
            time.sleep(0.001)
        os.close(wr)
    import threading
    import time
    th = threading.Thread(target=feed)
    th.start()
    r = Reader(rd, *opts)
    out = []
    while True:
        ps = r.next()
        if ps is None:
            break
        out.extend(zip(ps.seed_ids, ps.piles()))
    r.close()
    th.join()
    os.close(rd)
    assert out == _python(text, *opts)


def test_cli_native_path_orders_and_prints(tmp_path):
    """_run_native with a stand-in for the GPU: staging and finishing are called batch by
    batch, records leave in input order."""
    rng = random.Random(5)
    text = _rand_stream(rng, 30, with_noise=False)
    args = parse_args(["prog", "--min-n-read", "2", "--min-cov-aln", "0", "--output-full"])
    cfg = settings_from(args)
    want_piles = _python(text, 2, 0, 0, cfg.max_n_read, cfg.max_cov_aln)

    class FakeGpu:
        engines = [None]
        batch_bases = 900
        staged = 0

        def stage(self, ps):
            self.staged += 1
            return ps.piles()

        def finish(self, piles):
            return [(p[0] * 20)[:600] for p in piles]  # >= 500 chars so a record is printed

    path = tmp_path / "stream.txt"
    path.write_text(text)
    fd = os.open(str(path), os.O_RDONLY)
    out = io.StringIO()
    gpu = FakeGpu()
    try:
        _run_native(args, cfg, fd, gpu, out, batch_bases=900)
    finally:
        os.close(fd)
    want = "".join(">%s_f\n%s\n" % (sid, (p[0] * 20)[:600]) for sid, p in want_piles)
    assert out.getvalue() == want
    assert gpu.staged > 1

    # a failing output (closed pipe) ends the run with that error instead of hanging
    class Broken(io.StringIO):
        def write(self, _text):
            raise BrokenPipeError("stdout is gone")

    fd = os.open(str(path), os.O_RDONLY)
    try:
        with pytest.raises(BrokenPipeError):
            _run_native(args, cfg, fd, FakeGpu(), Broken(), batch_bases=900)
    finally:
        os.close(fd)

    # the devices are opened while the ingest thread is already reading (gpu given as a
    # factory): one that fails -- no HIP device -- ends the run with its error, the reader
    # is left and closed, nothing hangs; one that succeeds is used like the object itself
    def no_device():
        raise RuntimeError("no HIP device visible")

    for factory, error in ((no_device, RuntimeError), (FakeGpu, None)):
        fd = os.open(str(path), os.O_RDONLY)
        out = io.StringIO()
        try:
            if error is not None:
                with pytest.raises(error):
                    _run_native(args, cfg, fd, factory, out, batch_bases=900)
                assert out.getvalue() == ""
            else:
                _run_native(args, cfg, fd, factory, out, batch_bases=900)
                assert out.getvalue() == want
        finally:
            os.close(fd)


def test_bench_end_to_end_text_rebuilds_the_bench_piles(tmp_path):
    """bench.py's end-to-end leg writes its piles out as LA4Falcon text: the reader hands
    the worker exactly the piles the kernel-only legs run on."""
    import bench
    piles = bench.gen_piles([5, 6], 1, bench.WORKLOADS["ecoli"])
    path = tmp_path / "piles.txt"
    with open(path, "wb") as f:
        bench.write_la4falcon(piles, f)
    fd = os.open(path, os.O_RDONLY)
    try:
        r = Reader(fd, 10, 0, 10, 200, 0)
        ps = r.next()
        got = list(ps.piles())
        assert ps.seed_ids == ["%09d" % i for i in range(len(piles))]
        assert r.next() is None
        r.close()
    finally:
        os.close(fd)
    assert got == [[x.decode("ascii") for x in p] for p in piles]


def test_fuzzed_streams_match_the_python_parser():
    """Arbitrary ASCII byte soup -- control bytes, blank runs, empty and endless lines, marker
    look-alikes, no final newline -- parses like the python restatement, for several
    admission settings and batch limits (hypothesis; the reader is C++ over untrusted text)."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    token = st.one_of(
        st.sampled_from(["+", "*", "-", "+ +", "* *", "- -", "", " ", "\t", "\r", "\x00", "\x1c", "\x0b",
                         "r1", "r2", "r3", "seed", "x" * 40]),
        st.text(alphabet="ACGT", min_size=0, max_size=70),
        st.text(alphabet="ACGTN acgt\t\r\x00\x01\x1f+*-", min_size=0, max_size=30))
    line = st.lists(token, min_size=0, max_size=4).map(" ".join)
    stream = st.tuples(st.lists(line, min_size=0, max_size=60), st.booleans()).map(
        lambda t: "\n".join(t[0]) + ("\n" if t[1] else ""))

    @settings(max_examples=int(os.environ.get("FALCON_FUZZ_EXAMPLES", "500")), deadline=None, derandomize="FALCON_FUZZ_EXAMPLES" not in os.environ,
              suppress_health_check=list(HealthCheck))
    @given(stream, st.sampled_from([(1, 0, 0, 500, 0), (2, 3, 0, 4, 0), (3, 0, 1, 500, 2), (2, 0, 0, 3, 1)]),
           st.sampled_from([(0, 0), (1, 0), (0, 64), (3, 200)]))
    def check(text, opts, limits):
        # (seed ids leave the C ABI as C strings: a name holding a NUL byte ends there)
        want = [(sid.split("\x00")[0], pile) for sid, pile in _python(text, *opts)]
        got, _ = _native(text, *opts, max_piles=limits[0], max_bases=limits[1])
        assert got == want

    check()


def test_a_long_file_through_every_reading_mode(monkeypatch):
    """45 MB of stream -- several 4 MB pieces in flight at once, the text buffer growing under
    them -- read as a regular file (four pread() helpers, 64-byte compares where the host has
    AVX-512), with one helper, with the 16-byte scanner, and through a pipe: the same piles,
    and they are the stream's (consensus.py:161-209)."""
    rng = random.Random(4)
    base = _rand_seq(rng, 12000)
    lines, want = [], []
    for p in range(230):
        n_read = rng.randint(15, 25)
        names = ["%08d" % (1000 * p + i) for i in range(n_read)]
        seqs = [base[rng.randint(0, 500):rng.randint(9000, 12000)] for _ in range(n_read)]
        seqs[0] = base[:10000]
        for nm, sq in zip(names, seqs):
            lines.append("%s %s" % (nm, sq))
        lines.append("+ +")
        want.append((names[0], len(seqs[0]), n_read + 1))
    lines.append("- -")
    text = "\n".join(lines) + "\n"
    assert len(text) > 40e6
    ref, _ = _native(text, 1, 1, 0, 500, 0)
    assert [(sid, len(p[0]), len(p)) for sid, p in ref] == want
    monkeypatch.setenv("FALCON_AMD_READER_SLOTS1", "1")
    assert _native(text, 1, 1, 0, 500, 0)[0] == ref
    monkeypatch.setenv("FALCON_AMD_READER_SSE2", "1")
    assert _native(text, 1, 1, 0, 500, 0, max_bases=30_000_000)[0] == ref
    monkeypatch.delenv("FALCON_AMD_READER_SLOTS1")
    assert _native(text, 1, 1, 0, 500, 0, max_piles=7)[0] == ref
    monkeypatch.delenv("FALCON_AMD_READER_SSE2")
    # (the helpers list the white space of their pieces; the scanner reading every byte itself, as it did
    # through round 4, and other numbers of helpers)
    monkeypatch.setenv("FALCON_AMD_READER_SCAN_INLINE", "1")
    assert _native(text, 1, 1, 0, 500, 0, max_piles=7)[0] == ref
    monkeypatch.delenv("FALCON_AMD_READER_SCAN_INLINE")
    for n in ("2", "8"):
        monkeypatch.setenv("FALCON_AMD_READER_THREADS", n)
        assert _native(text, 1, 1, 0, 500, 0, max_bases=30_000_000)[0] == ref
    monkeypatch.delenv("FALCON_AMD_READER_THREADS")
    rd, wr = os.pipe()

    def feed():
        with os.fdopen(wr, "wb") as f:
            f.write(text.encode("ascii"))
    import threading
    th = threading.Thread(target=feed)
    th.start()
    r = Reader(rd, 1, 1, 0, 500, 0)
    got = []
    while True:
        ps = r.next(0, 0)
        if ps is None:
            break
        got.extend(zip(ps.seed_ids, ps.piles()))
    r.close()
    th.join()
    os.close(rd)
    assert got == ref



@pytest.mark.parametrize("ending", ["\r", "\r\n"])
def test_a_carriage_return_where_a_piece_of_the_file_ends(ending, monkeypatch):
    """The file is read in pieces of 4 MB whose white space the helper threads list: a line that ends in a
    lone '\\r' -- or whose '\\r\\n' is cut in two -- exactly where a piece ends is the one place where the
    scanner has to wait for the next piece (universal newlines: is the next byte a line feed?)."""
    rng = random.Random(5)
    piece = 4 << 20
    lines, size = [], 0
    p = 0
    while size < 3 * piece + 100000:
        n_read = 6
        pile = ["%08d %s" % (100 * p + i, _rand_seq(rng, rng.randint(3000, 5000))) for i in range(n_read)] + ["+ +"]
        for ln in pile:
            # the line that would cross the end of a piece is cut so that its '\r' is the piece's last byte
            nxt = (size // piece + 1) * piece
            if size + len(ln) + 1 > nxt and nxt - size > 40 and " " in ln and nxt - size < len(ln):
                ln = ln[:nxt - size - 1]
                lines.append(ln + ending)
                size += len(ln) + len(ending)
            else:
                lines.append(ln + "\n")
                size += len(ln) + 1
        p += 1
    lines.append("- -\n")
    text = "".join(lines)
    assert text[piece - 1] == "\r" and text[2 * piece - 1] == "\r"
    want = _python(text, 1, 1, 0, 500, 0)
    for limits in ((0, 0), (5, 0), (0, 200000)):
        got, _ = _native(text, 1, 1, 0, 500, 0, max_piles=limits[0], max_bases=limits[1])
        assert got == want
    monkeypatch.setenv("FALCON_AMD_READER_SCAN_INLINE", "1")
    assert _native(text, 1, 1, 0, 500, 0, max_piles=5)[0] == want
