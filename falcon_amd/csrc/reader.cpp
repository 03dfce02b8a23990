// reader.cpp -- native reader of the LA4Falcon "-fo" text stream.
//
// Host side of the consensus worker's ingest.  Restates, for whole
// batches and without an interpreter in the loop,
//   get_seq_data       falcon_kit/mains/consensus.py:161-209  (grammar, pile admission)
//   get_longest_reads  falcon_kit/mains/consensus.py:26-45    (read selection)
// and hands the selected piles to fa_batch_create() as plain pointer arrays.
//
// Grammar: a line that splits (on any run of white space) into exactly two tokens is
// "<name> <bases>"; "+ +" ends a pile, "* *" ends and discards it, "- -" ends the
// stream; every other line is skipped.  A sequence longer than 100000 is cut to 99999
// (:162,:176-177).  The first sequence of a pile is the seed: it is stored as the target
// and -- its name not having been seen yet -- once more as an ordinary read (:183-190).
// A pile is admitted when it holds >= min_n_read sequences and its read bases cover the
// seed >= min_cov_aln times (:196-198); admitted piles keep the seed plus the longest
// reads, stable on stream order, at most max_n_read sequences, optionally stopping once
// the depth exceeds max_cov_aln (:26-45).
//
// No HIP in this file: it is usable (and tested) without a GPU.
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <immintrin.h>
#include <unistd.h>

#include "../../include/falcon_amd.h"

namespace {

constexpr int kMaxSeqLen = 100000;  // consensus.py:162

struct Tok {
    size_t off;
    int len;
};

inline bool is_space(unsigned char c) {
    // str.split() without arguments: ASCII white space (the stream is ASCII)
    return c == ' ' || (c >= '\t' && c <= '\r') || (c >= 0x1c && c <= 0x1f);
}

}  // namespace

struct fa_reader {
    int fd = -1;
    int min_n_read = 0, min_len_aln = 0, min_cov_aln = 0, max_n_read = 0, max_cov_aln = 0;
    // time inside read() (helper thread), inside fa_reader_next, and of the latter waiting
    // for the former (FALCON_AMD_TIMING)
    double read_s = 0, next_s = 0, wait_s = 0, scan_s = 0;
    size_t read_bytes = 0;
    bool eof = false;        // read() returned 0
    bool finished = false;   // "- -" seen or eof reached and the buffer drained
    std::string err;

    // text of the batch being gathered; tokens are offsets into it until the batch is
    // closed (it may grow), pointers afterwards
    char *text = nullptr;    // (mmap'ed: grown by remapping, never copied or zero-filled by us)
    size_t text_cap = 0;
    // What a call hands out stays valid through the next n_keep - 1 calls (so that one thread
    // can stage a batch while another reads the following ones): n_keep text buffers and sets
    // of pointer arrays take turns (2 unless fa_reader_keep asked for more); `text` is the one
    // being filled, `shelf` holds the others between calls.
    struct Shelf {
        char *text = nullptr;
        size_t text_cap = 0;
        std::vector<int> pile_n_seq, out_len;
        std::vector<const char *> out_seqs, out_ids;
    };
    std::vector<Shelf> shelf = std::vector<Shelf>(2);
    bool handed_out = false; // fa_reader_next has been called (the ring keeps its size from then on)
    int cur = 0;             // shelf[cur]: arrays of the batch handed out last (its text is `text`,
                             // shelf[cur].text is null); shelf[cur + 1 (mod n_keep)]: the oldest one
    size_t parsed = 0;       // start of the line being scanned (all lines before it are split)
    size_t scanned = 0;      // bytes of `text` the scanner has looked at (>= parsed)
    size_t filled = 0;       // bytes of `text` holding stream data
    // white-space bytes (<= 0x20) seen so far in the line being scanned
    size_t line_low = 0, line_first_low = 0;
    ~fa_reader() {
        {
            std::lock_guard<std::mutex> g(ahead.mu);
            ahead.quit = true;
        }
        ahead.cv.notify_all();
        for (std::thread &th : ahead.th)
            if (th.joinable()) th.join();
        if (text) munmap(text, text_cap);
        for (const Shelf &o : shelf)
            if (o.text) munmap(o.text, o.text_cap);
    }

    // Read-ahead: while the scanner works through one 4 MB piece, helper threads have the
    // next ones under way (the copy out of the page cache or the pipe was 2/3 of the reader's
    // time, and one thread's copy -- ~11 GB/s -- is what bounded the whole worker in round 3).
    // A pipe has one helper and one read() at a time; a regular file has kSlots helpers, each
    // with a pread() of its own piece at its own offset.  Pieces are handed out and taken in
    // in stream order, always behind `filled`; the buffer only grows or changes hands while
    // no request is in flight.
    static constexpr int kSlots = 8;  // at most; n_slots of them are used
    struct Ahead {
        std::thread th[kSlots];
        std::mutex mu;
        std::condition_variable cv;
        struct Slot {
            enum { IDLE, REQUESTED, DONE } state = IDLE;
            char *dst = nullptr;
            size_t want = 0;
            long long off = -1;  // file offset (pread), -1: read() at the descriptor's position
            ssize_t n = 0;       // the result, -1 with `err` = errno
            int err = 0;
            // where the piece holds a byte <= 0x20 (offsets from dst, ascending): the helper looks
            // for them while the piece is in its cache, the scanner only walks the list.  `ev_ok`
            // false: no list (too many events to be LA4Falcon text) -- the scanner reads the bytes.
            std::vector<uint32_t> ev;
            bool ev_ok = false;
        } slot[kSlots];
        bool quit = false;
    } ahead;
    bool wide_scan = false;  // 64-byte compares (the host has AVX-512BW)
    bool scan_ahead = true;  // the helpers list the events of their pieces (FALCON_AMD_READER_SCAN_INLINE: no)
    int n_slots = 1;         // 1: a pipe; several: a regular file (file_off = where the next piece starts)
    // pieces taken in whose events are listed and which the scanner has not passed yet, in stream
    // order: [start, start + len) of `text` (start moves with the text when a batch's tail does)
    struct Piece {
        long long start = 0;
        size_t len = 0, next = 0;  // next: the first event not looked at yet
        std::vector<uint32_t> ev;
    };
    std::vector<Piece> pieces;      // (a handful: a vector used as a queue)
    std::vector<std::vector<uint32_t>> ev_spare;  // lists given back, for the slots to fill again
    long long file_off = -1;
    int in_flight = 0;       // requests out, slots (issue_at - in_flight .. issue_at - 1) mod n_slots
    int issue_at = 0;
    size_t pending = 0;      // bytes of the buffer behind `filled` that requests in flight will fill

    // pile in progress
    std::vector<Tok> pile;                  // seed, then reads in stream order
    std::unordered_set<std::string> names;  // read names seen in this pile
    Tok seed_name = {0, 0};
    long long pile_bases = 0;

    // closed piles of the batch
    std::vector<int> pile_n_seq;
    std::vector<Tok> sel;                   // selected sequences, pile after pile
    std::vector<Tok> sel_name;              // seed name per pile
    long long batch_bases = 0;

};

// The scanner's inner loop with 64-byte compares (AVX-512BW, chosen at run time: the GPU
// boxes' hosts have it, and one thread then looks at ~25 GB/s of text instead of 11): every
// byte <= 0x20 is an event.  Returns where it stopped (a multiple of 64 bytes behind `i`, or
// the event that closed the batch); the 16-byte loop takes what is left.
template <class F>
__attribute__((target("avx512f,avx512bw"))) static size_t scan64(const char *t, size_t i, size_t end, F &event,
                                                                 bool &go) {
    const __m512i lim = _mm512_set1_epi8(0x20);
    while (go && i + 64 <= end) {
        const __m512i v = _mm512_loadu_si512((const void *)(t + i));
        unsigned long long m = _mm512_cmple_epu8_mask(v, lim);
        for (; m; m &= m - 1)
            if (!event(i + (size_t)__builtin_ctzll(m))) {
                go = false;
                break;
            }
        i += 64;
    }
    return i;
}

// the same compares on a helper thread: where [t, t + n) holds a byte <= 0x20, as offsets.
// false (and no list) beyond kMaxEvents: text of one event per few bytes is not worth listing.
constexpr size_t kMaxEvents = 1u << 20;
__attribute__((target("avx512f,avx512bw"))) static size_t list64(const char *t, size_t n, std::vector<uint32_t> &out) {
    const __m512i lim = _mm512_set1_epi8(0x20);
    size_t i = 0;
    for (; i + 64 <= n && out.size() < kMaxEvents; i += 64) {
        unsigned long long m = _mm512_cmple_epu8_mask(_mm512_loadu_si512((const void *)(t + i)), lim);
        for (; m; m &= m - 1) out.push_back((uint32_t)(i + (size_t)__builtin_ctzll(m)));
    }
    return i;
}
static bool list_events(const char *t, size_t n, bool wide, std::vector<uint32_t> &out) {
    out.clear();
    size_t i = wide ? list64(t, n, out) : 0;
    const __m128i lim = _mm_set1_epi8(0x20);
    for (; i + 16 <= n && out.size() < kMaxEvents; i += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(t + i));
        unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(v, lim), v));
        for (; m; m &= m - 1) out.push_back((uint32_t)(i + (size_t)__builtin_ctz(m)));
    }
    for (; i < n && out.size() < kMaxEvents; i++)
        if ((unsigned char)t[i] <= 0x20) out.push_back((uint32_t)i);
    if (out.size() >= kMaxEvents) {
        out.clear();
        return false;
    }
    return true;
}

static char *map_text(char *old, size_t old_cap, size_t cap) {
    void *t = old ? mremap(old, old_cap, cap, MREMAP_MAYMOVE)
                  : mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (t == MAP_FAILED) return nullptr;
#ifdef MADV_HUGEPAGE
    (void)madvise(t, cap, MADV_HUGEPAGE);  // hundreds of MB touched once: fewer faults
#endif
    return (char *)t;
}

// a helper thread: serves its slot, one read() / pread() per request.  A pipe is polled first
// (in slices, looking at `quit`), so that closing the reader never waits for a producer that
// went silent.
static void ahead_main(fa_reader *r, int k) {
    fa_reader::Ahead &a = r->ahead;
    fa_reader::Ahead::Slot &sl = a.slot[k];
    std::unique_lock<std::mutex> lk(a.mu);
    for (;;) {
        a.cv.wait(lk, [&] { return a.quit || sl.state == fa_reader::Ahead::Slot::REQUESTED; });
        if (a.quit) return;
        char *dst = sl.dst;
        const size_t want = sl.want;
        const long long off = sl.off;
        lk.unlock();
        ssize_t n = -1;
        int err = 0;
        for (;;) {
            if (off < 0) {
                struct pollfd pfd = {r->fd, POLLIN, 0};
                const int pr = poll(&pfd, 1, 50);
                if (pr == 0 || (pr < 0 && errno == EINTR)) {
                    std::lock_guard<std::mutex> g(a.mu);
                    if (a.quit) return;
                    continue;
                }
                // (pr < 0 otherwise: not pollable -- just read)
            }
            const auto t0 = std::chrono::steady_clock::now();
            if (off < 0) {
                n = read(r->fd, dst, want);
            } else {  // a regular file: the whole piece, or what is left of the file
                size_t got = 0;
                n = 0;
                while (got < want) {
                    const ssize_t m = pread(r->fd, dst + got, want - got, (off_t)(off + (long long)got));
                    if (m < 0 && errno == EINTR) continue;
                    if (m < 0) { n = -1; break; }
                    if (m == 0) break;
                    got += (size_t)m;
                }
                if (n >= 0) n = (ssize_t)got;
            }
            err = errno;
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (n < 0 && (err == EINTR || err == EAGAIN)) continue;
            // (the slot's list is this thread's until the state says DONE)
            sl.ev_ok = n > 0 && r->scan_ahead && list_events(dst, (size_t)n, r->wide_scan, sl.ev);
            lk.lock();
            r->read_s += dt;
            break;
        }
        sl.n = n;
        sl.err = err;
        sl.state = fa_reader::Ahead::Slot::DONE;
        a.cv.notify_all();
    }
}

// ask for the next piece(s) of the stream, behind what is filled or already asked for
static bool request_ahead(fa_reader *r) {
    const size_t want = 4u << 20;
    fa_reader::Ahead &a = r->ahead;
    while (r->in_flight < r->n_slots) {
        if (r->text_cap < r->filled + r->pending + want) {
            if (r->in_flight > 0) break;  // (the buffer may move: not under a request's feet)
            const size_t need = r->filled + (size_t)r->n_slots * want;
            const size_t cap = (std::max(r->text_cap * 2, need) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            char *t = map_text(r->text, r->text_cap, cap);
            if (!t) {
                r->err = "falcon_amd: out of memory while reading the pile stream";
                r->eof = true;
                return false;
            }
            r->text = t;
            r->text_cap = cap;
        }
        const int k = r->issue_at;
        if (!a.th[k].joinable()) a.th[k] = std::thread(ahead_main, r, k);
        {
            std::lock_guard<std::mutex> g(a.mu);
            fa_reader::Ahead::Slot &sl = a.slot[k];
            sl.dst = r->text + r->filled + r->pending;
            if (sl.ev.capacity() == 0 && !r->ev_spare.empty()) {
                sl.ev.swap(r->ev_spare.back());
                r->ev_spare.pop_back();
            }
            sl.want = want;
            sl.off = r->file_off;
            sl.state = fa_reader::Ahead::Slot::REQUESTED;
        }
        if (r->file_off >= 0) r->file_off += (long long)want;
        r->pending += want;
        r->issue_at = (k + 1) % r->n_slots;
        r->in_flight++;
        a.cv.notify_all();
    }
    return true;
}

// wait for the OLDEST request in flight (if any) and take its bytes in; false: nothing came
// (end of file or an error, see r->eof / r->err)
static bool settle_ahead(fa_reader *r) {
    if (r->in_flight == 0) return true;
    fa_reader::Ahead &a = r->ahead;
    const int k = ((r->issue_at - r->in_flight) % r->n_slots + r->n_slots) % r->n_slots;
    r->in_flight--;
    ssize_t n;
    int err;
    size_t want;
    {
        std::unique_lock<std::mutex> lk(a.mu);
        fa_reader::Ahead::Slot &sl = a.slot[k];
        const auto t0 = std::chrono::steady_clock::now();
        a.cv.wait(lk, [&] { return sl.state == fa_reader::Ahead::Slot::DONE; });
        r->wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        n = sl.n;
        err = sl.err;
        want = sl.want;
        if (n > 0 && !r->eof && sl.ev_ok) {
            r->pieces.emplace_back();
            fa_reader::Piece &pc = r->pieces.back();
            pc.start = (long long)r->filled;
            pc.len = (size_t)n;
            pc.ev.swap(sl.ev);
        }
        sl.ev_ok = false;
        sl.state = fa_reader::Ahead::Slot::IDLE;
    }
    r->pending -= want;
    if (n > 0 && !r->eof) {
        r->filled += (size_t)n;
        r->read_bytes += (size_t)n;
        // (a piece of a file that came short is the file's last one: the pieces asked for
        // behind it start beyond the end and bring nothing)
        if ((size_t)n < want && r->file_off >= 0) r->eof = true;
        return true;
    }
    r->eof = true;
    if (n < 0) r->err = std::string("falcon_amd: read() of the pile stream failed: ") + strerror(err);
    return false;
}

// every request in flight, in order (the buffer is about to change hands)
static bool settle_all(fa_reader *r) {
    bool ok = true;
    while (r->in_flight > 0) ok = settle_ahead(r) && ok;
    return ok;
}

static bool fill(fa_reader *r) {
    // More of the stream behind `filled`; false at end of file.  Pieces are 4 MB so that what
    // is read is scanned while it is still in the cache, and so that the tail a closed batch
    // leaves behind (copied over by the next call) stays small.  The piece after the one
    // returned is requested before the scanner gets this one.
    if (r->eof) {
        (void)settle_all(r);
        return false;
    }
    if (r->in_flight == 0 && !request_ahead(r)) return false;
    const size_t before = r->filled;
    if (!settle_ahead(r) || r->filled == before) {
        (void)settle_all(r);
        return false;
    }
    if (r->eof) return true;  // (the file's last piece: nothing more to ask for)
    return request_ahead(r);
}

// get_longest_reads (consensus.py:26-45) on the pile in progress -> r->sel
static void close_pile(fa_reader *r) {
    std::vector<Tok> &p = r->pile;
    std::vector<Tok> rest(p.begin() + 1, p.end());
    std::stable_sort(rest.begin(), rest.end(), [](const Tok &a, const Tok &b) { return a.len > b.len; });
    size_t keep = (size_t)std::max(r->max_n_read, 0);
    if (r->max_cov_aln > 0) {
        const long long seed_len = p[0].len;
        long long depth_bases = 0;
        size_t k = 1;
        for (const Tok &t : rest) {
            if (depth_bases / seed_len > r->max_cov_aln) break;
            k++;
            depth_bases += t.len;
        }
        keep = std::min(k, keep);
    }
    const size_t n = std::min(keep, rest.size() + 1);
    if (n == 0) {  // max_n_read == 0: the reference would hand an empty list on; nothing to do
        return;
    }
    r->sel.push_back(p[0]);
    r->batch_bases += p[0].len;
    for (size_t i = 0; i + 1 < n; i++) {
        r->sel.push_back(rest[i]);
        r->batch_bases += rest[i].len;
    }
    r->pile_n_seq.push_back((int)n);
    r->sel_name.push_back(r->seed_name);
}

static void reset_pile(fa_reader *r) {
    r->pile.clear();
    r->names.clear();
    r->pile_bases = 0;
    r->seed_name = {0, 0};
}

// one line [b, e) of r->text (without its '\n') holding `n_low` bytes <= 0x20, the first of
// them at `first_low`; returns false when "- -" ends the stream
static bool take_line(fa_reader *r, size_t b, size_t e, size_t n_low, size_t first_low) {
    const char *t = r->text;
    Tok tok[2];
    int n_tok = 0;
    // the usual line is "<name> <bases>": one blank inside it, nothing else at or below
    // 0x20 (all white space is); anything else is tokenised byte by byte
    if (n_low == 1 && t[first_low] == ' ' && first_low != b && first_low + 1 < e) {
        tok[0] = {b, (int)(first_low - b)};
        tok[1] = {first_low + 1, (int)std::min<size_t>(e - (first_low + 1), 0x7fffffff)};
        n_tok = 2;
    } else if (n_low != 0) {
        size_t i = b;
        while (i < e) {
            while (i < e && is_space((unsigned char)t[i])) i++;
            if (i >= e) break;
            size_t j = i;
            while (j < e && !is_space((unsigned char)t[j])) j++;
            if (n_tok < 2) tok[n_tok] = {i, (int)std::min<size_t>(j - i, 0x7fffffff)};
            n_tok++;
            i = j;
        }
    }
    if (n_tok != 2) return true;
    Tok name = tok[0], seq = tok[1];
    if (seq.len > kMaxSeqLen) seq.len = kMaxSeqLen - 1;
    if (name.len == 1 && (t[name.off] == '+' || t[name.off] == '*' || t[name.off] == '-')) {
        if (t[name.off] == '-') return false;
        if (t[name.off] == '+' && !r->pile.empty() && (long long)r->pile.size() >= r->min_n_read &&
            r->pile_bases / r->pile[0].len >= r->min_cov_aln)
            close_pile(r);
        reset_pile(r);
        return true;
    }
    if (seq.len < r->min_len_aln) return true;
    if (r->pile.empty()) {
        r->pile.push_back(seq);
        r->seed_name = name;
    }
    if (r->names.emplace(t + name.off, (size_t)name.len).second) {
        r->pile.push_back(seq);
        r->pile_bases += seq.len;
    }
    return true;
}

extern "C" fa_reader *fa_reader_open(int fd, int min_n_read, int min_len_aln, int min_cov_aln,
                                     int max_n_read, int max_cov_aln) {
    fa_reader *r = new fa_reader;
    r->fd = fd;
    r->min_n_read = min_n_read;
    r->min_len_aln = min_len_aln;
    r->min_cov_aln = min_cov_aln;
    r->max_n_read = max_n_read;
    r->max_cov_aln = max_cov_aln;
#ifdef F_SETPIPE_SZ
    // a pipe from LA4Falcon: fewer, larger reads (refused for anything that is not a pipe)
    (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#endif
    r->wide_scan = __builtin_cpu_supports("avx512bw") && !getenv("FALCON_AMD_READER_SSE2");
    r->scan_ahead = !getenv("FALCON_AMD_READER_SCAN_INLINE");
    // a regular file (a block's LA4Falcon output kept on disk, the benchmarks): several
    // pread()s at a time from where the descriptor stands
    struct stat st;
    if (!getenv("FALCON_AMD_READER_SLOTS1") && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
        const off_t at = lseek(fd, 0, SEEK_CUR);
        if (at >= 0) {
            r->file_off = (long long)at;
            // (a helper copies its piece out of the page cache and lists its events: ~8 GB/s each)
            const char *want = getenv("FALCON_AMD_READER_THREADS");
            r->n_slots = std::max(1, std::min((int)fa_reader::kSlots, want ? atoi(want) : 6));
        }
    }
    return r;
}

extern "C" int fa_reader_keep(fa_reader *r, int n_batches) {
    if (!r || n_batches < 2 || n_batches > 64) return -1;
    if (r->handed_out) {
        r->err = "falcon_amd: fa_reader_keep after the first fa_reader_next";
        return -1;
    }
    r->shelf.resize((size_t)n_batches);
    return 0;
}

extern "C" void fa_reader_close(fa_reader *r) {
    if (r && getenv("FALCON_AMD_TIMING"))
        fprintf(stderr, "[falcon_amd] reader: %.1f MB, %.1f ms in fa_reader_next (%.1f ms of them waiting "
                "for data, %.1f ms scanning and splitting), %.1f ms in read()\n",
                r->read_bytes / 1e6, 1e3 * r->next_s, 1e3 * r->wait_s, 1e3 * r->scan_s, 1e3 * r->read_s);
    delete r;
}

extern "C" const char *fa_reader_error(const fa_reader *r) { return r ? r->err.c_str() : ""; }

extern "C" int fa_reader_next(fa_reader *r, int max_piles, long long max_bases, const int **pile_n_seq,
                              const char *const **seqs, const int **seq_len,
                              const char *const **seed_ids) {
    if (!r) return -1;
    r->handed_out = true;
    struct Clock {
        fa_reader *r;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Clock() { r->next_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } clock{r};
    if (r->finished) {  // (nothing is asked of the stream any more; a request in flight stays)
        r->pile_n_seq.clear();
        r->sel.clear();
        r->sel_name.clear();
        return r->err.empty() ? 0 : -1;
    }
    // pieces requested during the previous call land in the old buffer: take them along
    if (!settle_all(r) && !r->err.empty()) return -1;
    // The batch handed out last stays where it is (its pointers live through this call);
    // the unparsed tail and the lines of the pile in progress move over to the text buffer of
    // the OLDEST batch still on the shelves, and the stream continues there.
    {
        size_t keep_from = r->parsed;
        for (const Tok &t : r->pile) keep_from = std::min(keep_from, t.off);
        if (!r->pile.empty()) keep_from = std::min(keep_from, r->seed_name.off);
        // (names of the pile in progress are owned copies)
        if (keep_from > 0) {
            const size_t tail = r->filled - keep_from;
            const int nxt = (r->cur + 1) % (int)r->shelf.size();
            fa_reader::Shelf &mine = r->shelf[r->cur], &other = r->shelf[nxt];
            if (other.text_cap < tail) {
                const size_t cap = std::max(r->text_cap,  // (untouched pages cost nothing)
                                            (tail + (4u << 20) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
                char *t = map_text(other.text, other.text_cap, cap);
                if (!t) {
                    r->err = "falcon_amd: out of memory while reading the pile stream";
                    return -1;
                }
                other.text = t;
                other.text_cap = cap;
            }
            memcpy(other.text, r->text + keep_from, tail);
            mine.text = r->text;
            mine.text_cap = r->text_cap;
            r->text = other.text;
            r->text_cap = other.text_cap;
            other.text = nullptr;
            other.text_cap = 0;
            r->cur = nxt;
            r->filled -= keep_from;
            r->parsed -= keep_from;
            r->scanned -= keep_from;
            r->line_first_low -= std::min(r->line_first_low, keep_from);
            for (fa_reader::Piece &pc : r->pieces) pc.start -= (long long)keep_from;
            for (Tok &t : r->pile) t.off -= keep_from;
            if (!r->pile.empty()) r->seed_name.off -= keep_from;
        }
        r->pile_n_seq.clear();
        r->sel.clear();
        r->sel_name.clear();
        r->batch_bases = 0;
    }
    if (max_piles <= 0) max_piles = 0x7fffffff;
    if (max_bases <= 0) max_bases = 0x7fffffffffffffffll;
    // One pass over the text: every byte <= 0x20 is an event (16 bytes per compare); a
    // '\n' closes the line, anything else is counted as white space of the line.
    auto batch_open = [&]() {
        return !r->finished && (long long)r->pile_n_seq.size() < max_piles && r->batch_bases < max_bases;
    };
    // Line ends are python's universal newlines (a text-mode sys.stdin, which is what
    // consensus.py:170 iterates): '\n', '\r\n', and a lone '\r'.  A '\r' in front of a '\n'
    // is just white space of its line; the scanner never looks at a '\r' whose successor is
    // not in the buffer yet (see `end` below), so text[i + 1] exists here unless the stream
    // has ended.
    auto event = [&](size_t i) {  // false: stop scanning (batch closed or stream finished)
        const char c = r->text[i];
        if (c != '\n' && !(c == '\r' && (i + 1 >= r->filled || r->text[i + 1] != '\n'))) {
            if (r->line_low++ == 0) r->line_first_low = i;
            return true;
        }
        const size_t b = r->parsed, n_low = r->line_low, first_low = r->line_first_low;
        r->parsed = r->scanned = i + 1;
        r->line_low = 0;
        if (!take_line(r, b, i, n_low, first_low)) r->finished = true;
        return batch_open();
    };
    const __m128i lim = _mm_set1_epi8(0x20);
    while (batch_open()) {
        // a '\r' at the very end of what has been read waits for its successor
        size_t end = r->filled;
        if (!r->eof && end > r->scanned && r->text[end - 1] == '\r') end--;
        if (r->scanned >= end) {
            if (fill(r)) continue;
            if (!r->err.empty()) return -1;
            if (r->scanned < r->filled) continue;  // the '\r' that waited: the stream ends behind it
            // end of file: a last line without '\n' still counts (python iterates it too)
            if (r->parsed < r->filled) {
                const size_t b = r->parsed;
                r->parsed = r->filled;
                take_line(r, b, r->filled, r->line_low, r->line_first_low);
                r->line_low = 0;
            }
            r->finished = true;
            break;
        }
        const auto ts0 = std::chrono::steady_clock::now();
        const char *t = r->text;
        size_t i = r->scanned;
        bool go = true;
        // pieces the scanner is past give their lists back
        while (!r->pieces.empty() && r->pieces.front().start + (long long)r->pieces.front().len <= (long long)i) {
            r->ev_spare.emplace_back();
            r->ev_spare.back().swap(r->pieces.front().ev);
            r->pieces.erase(r->pieces.begin());
        }
        if (!r->pieces.empty()) {
            fa_reader::Piece &pc = r->pieces.front();
            if (pc.start == (long long)i || (pc.start < (long long)i && pc.next > 0)) {
                // the piece's events one by one (`next`: where an earlier visit stopped -- a '\r' at
                // the end of what had been read by then)
                const size_t base = (size_t)pc.start, stop = std::min(end, base + pc.len);
                size_t k = pc.next;
                for (; go && k < pc.ev.size(); k++) {
                    const size_t at = base + pc.ev[k];
                    if (at >= stop) break;
                    if (!event(at)) go = false;
                }
                pc.next = k;  // (the event that closed the batch is behind it: the loop stepped past it)
                if (go) r->scanned = stop;
                r->scan_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
                continue;
            }
            // (in the middle of a piece whose list was never begun -- the tail a closed batch left --
            // or in front of it: byte by byte as far as its end / its start)
            end = std::min(end, pc.start <= (long long)i ? (size_t)pc.start + pc.len : (size_t)pc.start);
        }
        if (r->wide_scan) i = scan64(t, i, end, event, go);
        while (go && i + 16 <= end) {
            const __m128i v = _mm_loadu_si128((const __m128i *)(t + i));
            unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(v, lim), v));
            for (; m; m &= m - 1)
                if (!event(i + (size_t)__builtin_ctz(m))) {
                    go = false;
                    break;
                }
            i += 16;
        }
        for (; go && i < end; i++)
            if ((unsigned char)t[i] <= 0x20 && !event(i)) go = false;
        if (go) r->scanned = end;  // (a stop left `scanned` just behind its line feed)
        r->scan_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
    }
    // the batch is closed: offsets become pointers, tokens become C strings
    const size_t n_sel = r->sel.size(), n_pile = r->pile_n_seq.size();
    char *t = r->text;
    fa_reader::Shelf &out = r->shelf[r->cur];  // (last used by the oldest batch, now given up)
    out.pile_n_seq = r->pile_n_seq;
    out.out_seqs.resize(n_sel);
    out.out_len.resize(n_sel);
    out.out_ids.resize(n_pile);
    for (size_t i = 0; i < n_sel; i++) {
        out.out_seqs[i] = t + r->sel[i].off;
        out.out_len[i] = r->sel[i].len;
    }
    for (size_t i = 0; i < n_pile; i++) {
        t[r->sel_name[i].off + (size_t)r->sel_name[i].len] = '\0';  // the separator after the name
        out.out_ids[i] = t + r->sel_name[i].off;
    }
    if (pile_n_seq) *pile_n_seq = out.pile_n_seq.data();
    if (seqs) *seqs = out.out_seqs.data();
    if (seq_len) *seq_len = out.out_len.data();
    if (seed_ids) *seed_ids = out.out_ids.data();
    return (int)n_pile;
}
