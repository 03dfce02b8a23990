// engine.hip -- host side of the batch ABI (include/falcon_amd.h section 2):
// staging, HBM layout, stage scheduling on one HIP stream, result fetch.
//
// Replaces the per-pile driver generate_consensus() (src/c/falcon.c:562-666) and
// the process pool around it (falcon_kit/mains/consensus.py:264-274): instead of
// one C call per pile per worker process, all piles of a batch go through each
// stage together.
#include "../../include/falcon_amd.h"
#include "fa_internal.h"
#include "fa_host.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;

// FALCON_AMD_TRACE=1: synchronise and report after every stage (debugging aid)
static bool trace_on() {
    static const bool v = getenv("FALCON_AMD_TRACE") != nullptr;  // (one thread-safe initialisation: several threads ask)
    return v;
}
static void trace_stage(hipStream_t s, const char *name) {
    if (!trace_on()) return;
    hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, "[falcon_amd] stage %-10s %s\n", name, hipGetErrorString(e));
    fflush(stderr);
}

// FALCON_AMD_TIMING=1: host-side wall time of the phases of fa_batch_create / fa_batch_submit
// on stderr (where do a worker's stalls come from?)
static bool timing_on() {
    static const bool v = getenv("FALCON_AMD_TIMING") != nullptr;  // (one thread-safe initialisation: several threads ask)
    return v;
}
struct PhaseTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string line;
    const char *what;
    explicit PhaseTimer(const char *w) : what(w) {}
    void mark(const char *name) {
        if (!timing_on()) return;
        auto t1 = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof(buf), " %s %.1f", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        line += buf;
        t0 = t1;
    }
    ~PhaseTimer() {
        if (timing_on() && !line.empty()) fprintf(stderr, "[falcon_amd] %s ms:%s\n", what, line.c_str());
    }
};

static void set_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

#define HIP_OK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                    __LINE__);                                                         \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

#define HIP_OK_P(call)                                                                 \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                    __LINE__);                                                         \
            return nullptr;                                                            \
        }                                                                              \
    } while (0)

struct fa_ctx {
    int device = 0;
    int n_cu = 0;
    size_t total_mem = 0;
    // Streams.  `stream` (front): seed index, k-mer chaining, alignment -- one batch at a
    // time (they share the alignment arena; front_mu).  `back_stream`: the MSA stage of a
    // batch (k_tags, k_links, k_score, k_backtrace: atomics at the memory side, then one
    // wavefront per pile walking its levels in sequence), which runs next to the NEXT
    // batch's front stages: its kernels are bound by other things than k_align's instruction
    // issue, and share the device with it better than with one another.
    // `dl_stream`: result download, so that a fetch never queues behind another batch's kernels.
    // Two back streams take turns: k_score lasts as long for 500 piles as for 3000 (one
    // wavefront per pile, each walking its own levels), so a worker's small batches would
    // otherwise queue up behind one another there while most wave slots are free.
    static constexpr int N_BACK_MAX = 4;
    int n_back = 2;  // (FALCON_AMD_NBACK: experiments)
    hipStream_t stream = nullptr;
    hipStream_t back_stream[N_BACK_MAX] = {};
    unsigned back_turn = 0;  // (back_mu)
    hipStream_t dl_stream = nullptr;
    std::mutex front_mu, fetch_mu;
    // The second half of a run -- the MSA plan (host; it needs the alignment summaries, i.e.
    // waits for k_align) and the MSA kernels -- is not the submitting thread's business: the
    // context's PLANNER thread takes the batches in submit order, waits for each one's
    // summaries and queues its MSA stage on a back stream.  fa_batch_submit therefore never
    // waits for the device, and the front stream never waits for the host.
    std::mutex back_mu;   // back_turn
    std::thread planner;
    std::mutex plan_mu;                  // plan_q, plan_stop
    std::condition_variable plan_cv;     // the planner's: work or stop
    std::condition_variable done_cv;     // the waiters': some batch's back_state settled
    std::deque<fa_batch *> plan_q;
    bool plan_stop = false;
    // repeat launches of alignments k_align2 handed back share arena2, from whichever stream
    std::mutex redo_mu;
    hipEvent_t ev_redo = nullptr;  // the last repeat launch (redo_mu)
    char *h_dl = nullptr;    // pinned landing buffer of the downloads (grow only; fetch_mu)
    size_t h_dl_cap = 0;
    // alignment work-slot arena (grow only)
    FaAlignArena arena = {};
    size_t arena_cells_bytes = 0, arena_rows_bytes = 0;  // rows and rowx have equal size
    // a few worst-case slots for the alignments that outgrow the usual ones (grow only)
    FaAlignArena arena2 = {};
    size_t arena2_cells_bytes = 0, arena2_rows_bytes = 0;
    // k_align2 (two alignments per wavefront): one tape ring per resident wavefront (grow only)
    FaAlign2Arena a2 = {};
    size_t a2_bytes = 0;
    // staging of a batch's ASCII (grow only, one batch at a time): pinned host buffer, its
    // device twin, and a stream of their own so that the upload and pack of batch i+1 run
    // next to the kernels of batch i instead of queueing behind them
    std::mutex stage_mu;
    hipStream_t up_stream = nullptr;
    uint8_t *h_stage = nullptr, *d_stage = nullptr;
    size_t h_stage_cap = 0, d_stage_cap = 0;
    int *d_first_bad = nullptr;  // k_pack: lowest sequence index holding a byte other than ACGT
    int *d_counter2 = nullptr;   // work counter of the repeat launches (arena2)
};

// Device and pinned-host blocks are recycled per device for the life of the process: a
// worker builds and frees a batch (two dozen buffers) every ~50 ms, and hipFree waits for
// the whole device while hipMalloc of released VRAM waits for the driver to wipe it.  A
// freed block goes to a size-ordered free list; a request takes the smallest block that
// fits without wasting more than half of it, else allocates (sizes rounded up so that the
// batches of one stream, which differ by a few per cent, find each other's blocks).
// fa_destroy of a device's last context releases its lists.
struct BlockCache {
    std::mutex mu;
    std::multimap<size_t, void *> dev_free, host_free;
    static size_t round_up(size_t bytes) {
        if (bytes < 4096) return 4096;
        size_t step = (size_t)1 << (63 - __builtin_clzll(bytes));  // largest power of two <= bytes
        step >>= 3;                                                 // eighth-octave steps
        return (bytes + step - 1) & ~(step - 1);
    }
    void *take(std::multimap<size_t, void *> &fl, size_t bytes, size_t *got) {
        std::lock_guard<std::mutex> hold(mu);
        auto it = fl.lower_bound(bytes);
        if (it != fl.end() && it->first <= bytes + bytes / 2 + 4096) {
            void *p = it->second;
            *got = it->first;
            fl.erase(it);
            return p;
        }
        return nullptr;
    }
    void give(std::multimap<size_t, void *> &fl, void *p, size_t bytes) {
        std::lock_guard<std::mutex> hold(mu);
        fl.emplace(bytes, p);
    }
    void drop_all() {
        std::lock_guard<std::mutex> hold(mu);
        for (auto &kv : dev_free) (void)hipFree(kv.second);
        for (auto &kv : host_free) (void)hipHostFree(kv.second);
        dev_free.clear();
        host_free.clear();
    }
};
static std::mutex g_cache_mu;
static std::map<int, BlockCache *> g_cache;   // per device
static std::map<int, int> g_ctx_count;
static BlockCache *cache_of_current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> hold(g_cache_mu);
    BlockCache *&c = g_cache[dev];
    if (!c) c = new BlockCache();
    return c;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0, bytes = 0;
    BlockCache *home = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }  // (error paths: whatever was allocated goes with its owner)
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        home = cache_of_current_device();
        const size_t want = count * sizeof(T);
        p = (T *)home->take(home->dev_free, want, &bytes);
        if (p) return 0;
        bytes = BlockCache::round_up(want);
        hipError_t e = hipMalloc((void **)&p, bytes);
        if (e != hipSuccess) {
            // the free lists may hold what is missing
            home->drop_all();
            e = hipMalloc((void **)&p, bytes);
        }
        if (e != hipSuccess) {
            set_err("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            p = nullptr;
            bytes = 0;
            return -1;
        }
        return 0;
    }
    void release() {
        if (p) home->give(home->dev_free, p, bytes);
        p = nullptr;
        n = bytes = 0;
    }
};

// Pinned host memory, grow only: D2H / H2D copies of it are true DMA transfers
// (a pageable std::vector is first staged by the runtime, at a fraction of the rate).
template <class T>
struct HostBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0, bytes = 0;
    BlockCache *home = nullptr;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { release(); }
    int resize(size_t count) {
        if (count > cap) {
            release();
            home = cache_of_current_device();
            const size_t want = std::max<size_t>(count, 1) * sizeof(T);
            p = (T *)home->take(home->host_free, want, &bytes);
            if (!p) {
                bytes = BlockCache::round_up(want);
                hipError_t e = hipHostMalloc((void **)&p, bytes, hipHostMallocDefault);
                if (e != hipSuccess) {
                    set_err("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
                    p = nullptr;
                    bytes = 0;
                    return -1;
                }
            }
            cap = bytes / sizeof(T);
        }
        n = count;
        return 0;
    }
    void release() {
        if (p) home->give(home->host_free, p, bytes);
        p = nullptr;
        n = cap = bytes = 0;
    }
    T *data() { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

struct fa_batch {
    fa_ctx *ctx = nullptr;
    int n_pile = 0, n_seq = 0;
    bool pair_mode = false;   // fa_align_pairs: ranges are forced, no k-mer stages
    int band = FA_BAND;
    std::vector<FaSeq> seq;
    std::vector<FaPile> pile;
    std::vector<int> order, chain_order;
    // k_align2's base indices are 32 bits: stretches of whole piles of < 2^28 packed words each, one launch per
    // stretch (one stretch for every batch a worker builds); empty when a single pile is larger than that
    struct A2Group { u32 word_base; int order_begin, order_count; };
    std::vector<A2Group> a2_groups;
    std::vector<u64> ascii_off, script_off, probe_off;
    u64 n_words = 0, ascii_bytes = 0, script_words = 0, probe_words = 0;
    int max_read_len = 0, max_seed_len = 0, max_rows = 0, max_bins = 4;
    long long sum_len = 0, sum_seed = 0;

    const uint8_t *ascii_dev = nullptr;  // the context's staging buffer, while the batch is built
    DevBuf<u64> d_ascii_off, d_script_off, d_probe_off, d_probe;
    DevBuf<u32> d_words, d_kidx, d_kpos, d_script;
    DevBuf<FaSeq> d_seq;
    DevBuf<FaPile> d_pile;
    DevBuf<int> d_order, d_chain_order, d_redo;
    // MSA stage (k_msa.hip)
    DevBuf<FaTagAln> d_ta;
    DevBuf<u32> d_acc_first, d_desc, d_links;
    DevBuf<int> d_tcov, d_seg_cnt, d_score_ovf, d_seg_pile, d_seg_t0, d_wide;
    DevBuf<u32> d_seg_base, d_seg_first;
    DevBuf<unsigned long long> d_bound;
    DevBuf<uint8_t> d_insb;
    DevBuf<u64> d_t_off, d_link_off, d_link_cap;
    DevBuf<FaTInfo> d_tinfo;
    DevBuf<uint16_t> d_lvl_nlink;
    DevBuf<FaScoreOut> d_score_out;
    DevBuf<FaRange> d_range;
    DevBuf<FaAln> d_aln;
    DevBuf<FaNode> d_nodes;
    DevBuf<char> d_out_seq;
    DevBuf<int> d_out_eqv;
    DevBuf<FaPileOut> d_pile_out;

    HostBuf<FaRange> h_range;
    HostBuf<FaAln> h_aln;
    HostBuf<FaPileOut> h_pile_out;
    HostBuf<FaTagAln> h_ta;
    std::vector<int> pile_err, pile_err_arg;  // per pile: 0 fine, else why it has no consensus
    std::vector<int> pile_fixed_err;          // ... decided when the batch was built (3: a byte other than ACGT)
    std::map<int, std::string> pile_err_msg;
    bool msa_static = false;  // seg lists and t_off (functions of the seed lengths) uploaded
    std::vector<int> h_out_eqv;
    std::vector<std::string> h_result;
    std::string fasta;  // fa_batch_fasta's text
    bool have_range = false, have_aln = false, fetched = false, fetched_eqv = false;
    u64 out_slots = 0;
    fa_stats stats = {};
    // events of the last run: 0..3 on the front stream (index, chain, align), 4..11 on the
    // back stream (MSA stage), 12: alignment summaries on the host, 13 / 14: around the repeat
    // launch of the alignments k_align2 handed back (redo_timed: there was one in this run)
    hipEvent_t ev[15] = {};
    bool redo_timed = false;
    bool in_flight = false;  // fa_batch_submit done, fa_batch_wait pending
    // kernels of this batch were launched on the context's front stream (submit, pair and
    // unitig runs, also ones that failed half-way): fa_batch_free waits for that stream
    // before the batch's blocks go back to the cache, where another thread may take them
    bool front_launched = false;
    // The MSA stage (plan on the host, then k_tags .. k_backtrace on a back stream) is the
    // second half of a run, done by whichever thread gets to it first -- the next submit on
    // the context, or this batch's own wait: 0 not begun, 1 some thread is at it, 2 queued
    // on the back stream, -1 failed (back_err says why).
    std::atomic<int> back_state{0};
    std::string back_err;
    int back_rc = 0;
    struct {  // what the second half needs from the call that started the run
        unsigned min_cov = 0;
        double max_diff = 0;
        int band = 0, force_accept_g = -1;
        bool two_per_wave = false;
        int score_mode = 0, force_generic = 0;  // FALCON_AMD_SCORE1 / _SCORE_GENERIC, read by the calling thread
        int links_mode = 0;                     // FALCON_AMD_LINKS1
        size_t n_seg = 0;
        u64 t_tot = 0;
    } run;
    size_t n_ta = 0;  // accepted alignments of the last run's plan (h_ta)
    HostBuf<unsigned long long> h_a2_stats;
    HostBuf<u32> h_acc_first;
    HostBuf<u64> h_link_off, h_link_cap;
    HostBuf<FaPile> h_pile_up;
    ~fa_batch() {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    int ensure_events() {
        for (auto &e : ev)
            if (!e && hipEventCreate(&e) != hipSuccess) return -1;
        return 0;
    }

    const int *order_dev() const { return d_order.p; }
    FaBatchDev dev() const {
        FaBatchDev b;
        b.ascii = ascii_dev; b.ascii_off = d_ascii_off.p; b.words = d_words.p;
        b.seq = d_seq.p; b.pile = d_pile.p; b.n_seq = n_seq; b.n_pile = n_pile;
        b.n_words = n_words; b.kidx = d_kidx.p; b.kpos = d_kpos.p; b.order = d_order.p;
        b.chain_order = d_chain_order.p; b.n_chain = (int)chain_order.size();
        b.range = d_range.p; b.aln = d_aln.p; b.probe = d_probe.p; b.probe_off = d_probe_off.p;
        b.script = d_script.p; b.script_off = d_script_off.p; b.nodes = d_nodes.p;
        b.out_seq = d_out_seq.p; b.out_eqv = d_out_eqv.p; b.pile_out = d_pile_out.p;
        return b;
    }
};

extern "C" const char *fa_last_error(void) { return g_err.c_str(); }

extern "C" int fa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void planner_main(fa_ctx *c);

extern "C" fa_ctx *fa_create(int device) {
    PhaseTimer pt("fa_create");
    int n = fa_device_count();
    pt.mark("device-count");
    if (n <= 0) {
        set_err("falcon_amd: no HIP device visible -- this library has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_err("falcon_amd: device %d out of range (%d visible)", device, n);
        return nullptr;
    }
    HIP_OK_P(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OK_P(hipGetDeviceProperties(&prop, device));
    pt.mark("set-device+properties");
    fa_ctx *c = new fa_ctx();
    c->device = device;
    {
        std::lock_guard<std::mutex> hold(g_cache_mu);
        g_ctx_count[device]++;
    }
    c->n_cu = prop.multiProcessorCount;
    c->total_mem = prop.totalGlobalMem;
    HIP_OK_P(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {   // the back stream's few wavefronts go first when wave slots free up
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (const char *e = getenv("FALCON_AMD_NBACK")) c->n_back = std::max(1, std::min((int)fa_ctx::N_BACK_MAX, atoi(e)));
        const char *pr = getenv("FALCON_AMD_BACK_PRIO");  // (experiments: "low" / "same"; default: the highest)
        const int prio = pr && !strcmp(pr, "low") ? least : pr && !strcmp(pr, "same") ? 0 : greatest;
        for (int i = 0; i < c->n_back; i++)
            HIP_OK_P(hipStreamCreateWithPriority(&c->back_stream[i], hipStreamNonBlocking, prio));
    }
    HIP_OK_P(hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
    HIP_OK_P(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    pt.mark("streams");
    HIP_OK_P(hipMalloc((void **)&c->arena.counter, sizeof(int)));
    HIP_OK_P(hipMalloc((void **)&c->d_first_bad, sizeof(int)));
    HIP_OK_P(hipMalloc((void **)&c->d_counter2, sizeof(int)));
    HIP_OK_P(hipEventCreateWithFlags(&c->ev_redo, hipEventDisableTiming));
    HIP_OK_P(hipEventRecord(c->ev_redo, c->stream));
    HIP_OK_P(hipMalloc((void **)&c->arena.prof, 8 * sizeof(u64)));
    HIP_OK_P(hipMemset(c->arena.prof, 0, 8 * sizeof(u64)));
    HIP_OK_P(hipMalloc((void **)&c->a2.stats, 12 * sizeof(unsigned long long)));
    HIP_OK_P(hipMemset(c->a2.stats, 0, 12 * sizeof(unsigned long long)));
    c->a2.counter = c->arena.counter;  // (the front stream runs one alignment launch at a time)
    c->planner = std::thread(planner_main, c);
    pt.mark("small-buffers");
    return c;
}

static int stage_reserve(fa_ctx *c, size_t bytes, bool device_twin);

// What the first batch of a context would otherwise pay for on the spot: the pinned staging
// buffer for batches of `batch_bases` bases, the code objects of the path's kernels, the first
// host-to-device copy.  The workers call it from the thread that opens the engine, while the first
// batches are being read (round 3's first batch waited 25 ms for the pinned buffer, 30 ms for its
// upload and 39 ms for its first launches).
extern "C" int fa_warm(fa_ctx *c, long long batch_bases) {
    if (!c) return -1;
    PhaseTimer pt("fa_warm");
    HIP_OK(hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> hold(c->stage_mu);
        const size_t bytes = (size_t)std::max<long long>(batch_bases, 1 << 20) / 4 + (4u << 20);
        if (stage_reserve(c, bytes, false)) return -1;
        pt.mark("pinned-staging");
        HIP_OK(hipMemsetAsync(c->d_first_bad, 0, sizeof(int), c->up_stream));
        HIP_OK(hipMemcpyAsync(c->d_first_bad, c->h_stage, sizeof(int), hipMemcpyHostToDevice, c->up_stream));
        HIP_OK(hipStreamSynchronize(c->up_stream));
        pt.mark("first-copy");
    }
    fa_touch_index(); fa_touch_chain(); fa_touch_align2(); fa_touch_msa(); fa_touch_links2();
    fa_touch_score1(); fa_touch_score2();
    pt.mark("code-objects");
    return 0;
}

extern "C" void fa_destroy(fa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->planner.joinable()) {  // (it finishes what was submitted first)
        {
            std::lock_guard<std::mutex> hold(c->plan_mu);
            c->plan_stop = true;
        }
        c->plan_cv.notify_all();
        c->planner.join();
    }
    if (c->arena.cells) (void)hipFree(c->arena.cells);
    if (c->arena.rows) (void)hipFree(c->arena.rows);
    if (c->arena.rowx) (void)hipFree(c->arena.rowx);
    if (c->arena.counter) (void)hipFree(c->arena.counter);
    if (c->arena2.cells) (void)hipFree(c->arena2.cells);
    if (c->arena2.rows) (void)hipFree(c->arena2.rows);
    if (c->arena2.rowx) (void)hipFree(c->arena2.rowx);
    if (c->a2.mem) (void)hipFree(c->a2.mem);
    if (c->a2.stats) (void)hipFree(c->a2.stats);
    if (c->d_first_bad) (void)hipFree(c->d_first_bad);
    if (c->d_counter2) (void)hipFree(c->d_counter2);
    if (c->ev_redo) (void)hipEventDestroy(c->ev_redo);
    if (c->h_dl) (void)hipHostFree(c->h_dl);
    for (hipStream_t sb : c->back_stream)
        if (sb) (void)hipStreamDestroy(sb);
    if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    {   // the device's last context takes the recycled blocks with it
        BlockCache *bc = nullptr;
        {
            std::lock_guard<std::mutex> hold(g_cache_mu);
            if (--g_ctx_count[c->device] <= 0) {
                auto it = g_cache.find(c->device);
                if (it != g_cache.end()) bc = it->second;
            }
        }
        if (bc) bc->drop_all();
    }
    delete c;
}

// rows an alignment of this read can use: max_d = (int)(0.3*(dq+dt))
// (DW_banded.c:149) with dq <= len, dt <= T and, past the sanity filter of
// falcon.c:613-619, |dq-dt| <= 0.05*(dq+dt)  =>  dq+dt <= 2.106*len.
static int rows_bound(int len, int T, bool filtered) {
    double s = (double)len + (double)T;
    if (filtered) s = std::min(s, 2.106 * (double)len + 2.0);
    return (int)(0.3 * s) + 2;
}

// Grow the context's staging pair (pinned host buffer, device twin) to `bytes`.
// Caller holds stage_mu.
static int stage_reserve(fa_ctx *c, size_t bytes, bool device_twin) {
    if (bytes > c->h_stage_cap) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_cap = 0;
        const size_t cap = bytes + bytes / 8;  // batches of one run differ by a few per cent
        if (hipHostMalloc((void **)&c->h_stage, cap, hipHostMallocDefault) != hipSuccess) {
            set_err("falcon_amd: hipHostMalloc(%zu) failed", cap);
            c->h_stage = nullptr;
            return -1;
        }
        c->h_stage_cap = cap;
    }
    if (device_twin && bytes > c->d_stage_cap) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr;
        c->d_stage_cap = 0;
        const size_t cap = bytes + bytes / 8;
        if (hipMalloc((void **)&c->d_stage, cap) != hipSuccess) {
            set_err("falcon_amd: hipMalloc(%zu) failed", cap);
            c->d_stage = nullptr;
            return -1;
        }
        c->d_stage_cap = cap;
    }
    return 0;
}

// Threads that pack a batch into the staging buffer: one per 32 MB, at most 16 (or
// FALCON_AMD_STAGE_THREADS: a node of 8 GPUs has 16 cores per GPU), never more than the cores
// there are.
static int stage_threads(u64 bytes) {
    int cap = 16;
    if (const char *e = getenv("FALCON_AMD_STAGE_THREADS")) cap = std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc > 0) cap = std::min<int>(cap, (int)hc);
    return (int)std::max<u64>(1, std::min<u64>((u64)cap, bytes >> 25));
}

static fa_batch *batch_build(fa_ctx *ctx, int n_pile, const int *pile_n_seq,
                             const char *const *seqs, const int *seq_len, bool pair_mode,
                             int band) {
    if (!ctx) {
        set_err("falcon_amd: null context");
        return nullptr;
    }
    HIP_OK_P(hipSetDevice(ctx->device));
    PhaseTimer pt("fa_batch_create");
    fa_batch *b = new fa_batch();
    b->ctx = ctx;
    b->n_pile = n_pile;
    b->pair_mode = pair_mode;
    b->band = band;
    int g = 0;
    u64 woff = 0, aoff = 0, kpos = 0, out = 0;
    b->pile.resize(n_pile);
    for (int p = 0; p < n_pile; p++) {
        FaPile &pm = b->pile[p];
        memset(&pm, 0, sizeof(pm));
        pm.first = g;
        pm.n_seq = pile_n_seq[p];
        if (pm.n_seq < 1) {
            set_err("falcon_amd: pile %d has no seed", p);
            delete b;
            return nullptr;
        }
        for (int j = 0; j < pm.n_seq; j++, g++) {
            int len = seq_len ? seq_len[g] : (int)strlen(seqs[g]);
            if (len < 0 || len >= 100000 * 10) {
                set_err("falcon_amd: sequence %d has unsupported length %d", g, len);
                delete b;
                return nullptr;
            }
            FaSeq s;
            s.woff = (u32)woff;
            s.len = len;
            s.pile = p;
            s.idx = j;
            b->seq.push_back(s);
            b->ascii_off.push_back(aoff);
            u64 nw = (u64)((len + 15) / 16) + 2;
            nw = (nw + 3) & ~(u64)3;
            woff += nw;
            aoff += (((u64)len + 15) & ~(u64)15) + 16;
            b->sum_len += len;
            if (j == 0) {
                pm.seed_len = len;
                b->sum_seed += len;
                b->max_seed_len = std::max(b->max_seed_len, len);
            } else {
                b->max_read_len = std::max(b->max_read_len, len);
            }
        }
        // the reference asserts t_len < 100000 (falcon.c:343)
        if (!pair_mode && pm.seed_len >= 100000) {
            set_err("falcon_amd: seed of pile %d is %d bp; the reference asserts < 100000", p,
                    pm.seed_len);
            delete b;
            return nullptr;
        }
        pm.kidx_off = (u64)p * FA_IDX_STRIDE;
        pm.kpos_off = kpos;
        kpos += (u64)pm.seed_len + 4;
        pm.out_off = out;
        out += 2 * (u64)pm.seed_len + 4;
    }
    if (woff >= 0xffffffffull) {
        set_err("falcon_amd: batch too large (%llu packed words)", (unsigned long long)woff);
        delete b;
        return nullptr;
    }
    b->pile_err.assign(n_pile, 0);
    b->pile_err_arg.assign(n_pile, 0);
    b->pile_fixed_err.assign(n_pile, 0);
    b->n_seq = g;
    b->n_words = woff;
    b->ascii_bytes = aoff + 16;
    b->out_slots = out;
    // per-sequence scratch extents
    b->script_off.resize(g);
    b->probe_off.resize(g);
    u64 so = 0, po = 0;
    int max_rows = 4;
    for (int i = 0; i < g; i++) {
        const FaSeq &s = b->seq[i];
        int T = b->pile[s.pile].seed_len;
        b->script_off[i] = so;
        b->probe_off[i] = po;
        if (s.idx == 0) continue;
        po += (u64)(s.len / 4 + 4);
        // diagonals q-t span at most len+T, binned by K*6 (kmer_lookup.c:350-355)
        b->max_bins = std::max(b->max_bins, (s.len + T) / (FA_K * 6) + 4);
        int rb = rows_bound(s.len, T, !pair_mode);
        so += (u64)((rb + 3) & ~3);
        max_rows = std::max(max_rows, rb);
    }
    b->script_words = so;
    b->probe_words = po;
    b->max_rows = max_rows;
    // longest reads first: the lanes of a k_chain wave and the tail of the
    // k_align work queue then see similar work
    // (... inside a stretch of piles k_align2 takes in one launch: one stretch, unless the batch holds 2^28
    // packed words = 4.29 G bases or more -- FALCON_AMD_A2_MAX_WORDS: tests force the cut)
    std::vector<int> group_of(n_pile, 0);
    {
        u64 lim = (1ull << 28) - 64;
        if (const char *e = getenv("FALCON_AMD_A2_MAX_WORDS")) lim = std::max<u64>(64, strtoull(e, nullptr, 10));
        std::vector<u64> base;  // first packed word of every stretch
        bool ok = true;
        for (int p = 0; p < n_pile; p++) {
            const u64 first = b->seq[b->pile[p].first].woff;
            const u64 end = p + 1 < n_pile ? (u64)b->seq[b->pile[p + 1].first].woff : woff;
            if (base.empty() || end - base.back() > lim) base.push_back(first);
            if (end - base.back() > (1ull << 28) - 64) ok = false;   // one pile alone is too large: k_align takes the batch
            group_of[p] = (int)base.size() - 1;
        }
        b->a2_groups.clear();
        if (ok)
            for (u64 w : base) b->a2_groups.push_back({(u32)w, 0, 0});
    }
    b->order.resize(g);
    std::iota(b->order.begin(), b->order.end(), 0);
    std::stable_sort(b->order.begin(), b->order.end(), [&](int x, int y) {
        const int gx = group_of[b->seq[x].pile], gy = group_of[b->seq[y].pile];
        if (gx != gy) return gx < gy;
        int lx = b->seq[x].idx == 0 ? -1 : b->seq[x].len;
        int ly = b->seq[y].idx == 0 ? -1 : b->seq[y].len;
        return lx > ly;
    });
    for (int i = 0; i < g && !b->a2_groups.empty(); i++) {
        auto &gr = b->a2_groups[group_of[b->seq[b->order[i]].pile]];
        if (gr.order_count++ == 0) gr.order_begin = i;
    }
    // k_chain gathers from its pile's k-mer tables (262 KB + 4 B per seed base): its
    // work list is pile-major so that a pile's reads run together, and dealt into 8
    // interleaved streams (pile p -> stream p mod 8, workgroup b takes entry b / 8 of
    // stream b mod 8) so that -- with workgroups dealt round-robin to the 8 XCDs -- a
    // pile's tables live in ONE XCD's L2 (4 MB holds the ~7 piles an XCD works on).
    {
        std::vector<int> stream[8];
        for (int p = 0; p < n_pile; p++)
            for (int j = 0; j < b->pile[p].n_seq; j++) stream[p & 7].push_back(b->pile[p].first + j);
        size_t longest = 0;
        for (auto &v : stream) longest = std::max(longest, v.size());
        b->chain_order.assign(longest * 8, -1);
        for (int x = 0; x < 8; x++)
            for (size_t j = 0; j < stream[x].size(); j++) b->chain_order[j * 8 + x] = stream[x][j];
    }

    // The sequences are packed to 2 bits per base on the way into the context's pinned
    // buffer, by a few threads (pack_host.cpp), and uploaded as they will lie in HBM: a
    // quarter of the host writes and of the PCIe bytes of the text, which the device never
    // sees.  (FALCON_AMD_DEVICE_PACK=1: the text is copied, uploaded and packed by k_pack --
    // round 2's route, kept for the counter calibration of scripts/pmc_traffic_record.py.)
    pt.mark("layout");
    static const bool device_pack = getenv("FALCON_AMD_DEVICE_PACK") != nullptr;
    int rc = 0;
    if (device_pack) rc |= b->d_ascii_off.alloc(g);
    // (k_align2 fills its LDS windows with 256 words from wherever a band stands: room behind the
    // last sequence for a window that begins at its last word)
    rc |= b->d_words.alloc(b->n_words + 8 + 320);
    rc |= b->d_seq.alloc(g);
    rc |= b->d_pile.alloc(n_pile);
    rc |= b->d_order.alloc(g);
    rc |= b->d_chain_order.alloc(b->chain_order.size() + 1);
    rc |= b->d_range.alloc(g);
    rc |= b->d_aln.alloc(g);
    rc |= b->d_script_off.alloc(g);
    rc |= b->d_script.alloc(b->script_words + 8);
    if (!pair_mode) {
        rc |= b->d_kidx.alloc((u64)n_pile * FA_IDX_STRIDE);
        rc |= b->d_kpos.alloc(kpos + 8);
        rc |= b->d_probe_off.alloc(g);
        rc |= b->d_probe.alloc(b->probe_words + 8);
        rc |= b->d_out_seq.alloc(b->out_slots + 8);
        rc |= b->d_out_eqv.alloc(b->out_slots + 8);
        rc |= b->d_pile_out.alloc(n_pile);
    }
    if (rc) {
        delete b;
        return nullptr;
    }
    bool ok = true;
    pt.mark("device-buffers");
    {
        std::lock_guard<std::mutex> hold(ctx->stage_mu);
        pt.mark("stage-lock");
        const size_t stage_bytes = device_pack ? (size_t)b->ascii_bytes : (size_t)b->n_words * sizeof(u32);
        if (stage_reserve(ctx, stage_bytes, device_pack)) {
            delete b;
            return nullptr;
        }
        pt.mark("stage-reserve");
        uint8_t *h_ascii = ctx->h_stage;
        u32 *h_words = reinterpret_cast<u32 *>(ctx->h_stage);
        const u64 *aoffs = b->ascii_off.data();
        const FaSeq *sq = b->seq.data();
        const u64 n_words = b->n_words;
        std::vector<int> bad_at((size_t)g, -1);  // per sequence: its first byte outside ACGT
        int *bad_at_p = bad_at.data();
        auto copy_part = [=](int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                if (device_pack) {
                    memcpy(h_ascii + aoffs[i], seqs[i], (size_t)sq[i].len);
                } else {
                    const u64 end = i + 1 < g ? (u64)sq[i + 1].woff : n_words;
                    bad_at_p[i] = fa_pack_host(seqs[i], sq[i].len, h_words + sq[i].woff, (long long)(end - sq[i].woff));
                }
            }
        };
        const int n_thr = stage_threads(b->ascii_bytes);
        if (n_thr <= 1) {
            copy_part(0, g);
        } else {
            // equal shares of the bytes: sequence ranges cut where ascii_off crosses k/n
            std::vector<std::thread> pool;
            int i0 = 0;
            for (int k = 1; k <= n_thr; k++) {
                const u64 cut = b->ascii_bytes / (u64)n_thr * (u64)k;
                int i1 = k == n_thr ? g
                                    : (int)(std::lower_bound(aoffs + i0, aoffs + g, cut) - aoffs);
                if (i1 > i0) pool.emplace_back(copy_part, i0, i1);
                i0 = i1;
            }
            for (auto &t : pool) t.join();
        }
        pt.mark(device_pack ? "host-copy" : "host-pack");
        hipStream_t s = ctx->up_stream;
        if (device_pack) {
            b->ascii_dev = ctx->d_stage;
            ok &= hipMemcpyAsync(ctx->d_stage, h_ascii, b->ascii_bytes, hipMemcpyHostToDevice, s) == hipSuccess;
            ok &= hipMemcpyAsync(b->d_ascii_off.p, b->ascii_off.data(), g * sizeof(u64), hipMemcpyHostToDevice, s) == hipSuccess;
        } else {
            ok &= n_words == 0 ||
                  hipMemcpyAsync(b->d_words.p, h_words, (size_t)n_words * sizeof(u32), hipMemcpyHostToDevice, s) == hipSuccess;
        }
        ok &= hipMemcpyAsync(b->d_seq.p, b->seq.data(), g * sizeof(FaSeq), hipMemcpyHostToDevice, s) == hipSuccess;
        ok &= hipMemcpyAsync(b->d_pile.p, b->pile.data(), n_pile * sizeof(FaPile), hipMemcpyHostToDevice, s) == hipSuccess;
        ok &= hipMemcpyAsync(b->d_order.p, b->order.data(), g * sizeof(int), hipMemcpyHostToDevice, s) == hipSuccess;
        ok &= b->chain_order.empty() ||
              hipMemcpyAsync(b->d_chain_order.p, b->chain_order.data(), b->chain_order.size() * sizeof(int),
                             hipMemcpyHostToDevice, s) == hipSuccess;
        ok &= hipMemcpyAsync(b->d_script_off.p, b->script_off.data(), g * sizeof(u64), hipMemcpyHostToDevice, s) == hipSuccess;
        if (!pair_mode)
            ok &= hipMemcpyAsync(b->d_probe_off.p, b->probe_off.data(), g * sizeof(u64), hipMemcpyHostToDevice, s) == hipSuccess;
        trace_stage(s, "upload");
        int first_bad = 0x7fffffff;
        std::vector<int> bad((size_t)n_pile, 0);  // piles holding a byte outside ACGT
        DevBuf<int> d_bad_pile;
        if (device_pack) {
            if (ok && !pair_mode) {
                ok &= d_bad_pile.alloc((size_t)n_pile) == 0;
                if (ok) ok &= hipMemsetAsync(d_bad_pile.p, 0, (size_t)n_pile * sizeof(int), s) == hipSuccess;
            }
            if (ok) {
                ok &= hipMemcpyAsync(ctx->d_first_bad, &first_bad, sizeof(int), hipMemcpyHostToDevice, s) == hipSuccess;
                fa_launch_pack(b->dev(), ctx->d_first_bad, pair_mode ? nullptr : d_bad_pile.p, s);
                ok &= hipGetLastError() == hipSuccess;
                ok &= hipMemcpyAsync(&first_bad, ctx->d_first_bad, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess;
            }
        } else {
            for (int i = 0; i < g; i++) {
                if (bad_at[i] < 0) continue;
                first_bad = std::min(first_bad, i);
                bad[b->seq[i].pile] = 1;
            }
        }
        // (also on the error path: nothing may still read the staging buffers)
        ok &= hipStreamSynchronize(s) == hipSuccess;
        trace_stage(s, "pack");
        pt.mark(device_pack ? "h2d+pack" : "h2d");
        b->ascii_dev = nullptr;
        if (device_pack && ok && first_bad != 0x7fffffff && !pair_mode)
            ok &= hipMemcpy(bad.data(), d_bad_pile.p, (size_t)n_pile * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
        auto describe = [&](int g) {
            // the reference aligns raw characters and codes other bytes specially
            // (kmer_lookup.c:159-171, :236-249): outside the parity domain
            const FaSeq &sq = b->seq[g];
            int at = 0;
            while (at < sq.len && seqs[g][at] && strchr("ACGT", seqs[g][at])) at++;
            char msg[256];
            snprintf(msg, sizeof(msg), "sequence %d of pile %d holds byte 0x%02x at position %d; only "
                     "upper-case A, C, G, T are supported (what LA4Falcon emits)", sq.idx, sq.pile,
                     at < sq.len ? (unsigned)(unsigned char)seqs[g][at] : 0u, at);
            return std::string(msg);
        };
        if (ok && first_bad != 0x7fffffff && pair_mode) {
            // (the legacy align() has no way to report one pair: the call fails)
            set_err("falcon_amd: %s", describe(first_bad).c_str());
            delete b;
            return nullptr;
        }
        if (ok && first_bad != 0x7fffffff) {
            // Piles fail alone: a pile with such a sequence gets no consensus and an entry in
            // fa_batch_pile_error; its reads are taken out of the stages (as index 0 they
            // count as seeds: no chain, no alignment); the batch -- and the stream -- go on.
            for (int p = 0; ok && p < n_pile; p++) {
                if (!bad[p]) continue;
                const FaPile &pm = b->pile[p];
                int g_bad = pm.first;
                for (int j = 0; j < pm.n_seq; j++) {
                    const int g = pm.first + j;
                    int at = 0;
                    while (at < b->seq[g].len && seqs[g][at] && strchr("ACGT", seqs[g][at])) at++;
                    if (at < b->seq[g].len) { g_bad = g; break; }
                }
                b->pile_fixed_err[p] = 3;
                b->pile_err_msg[p] = describe(g_bad);
                for (int j = 0; j < pm.n_seq; j++) b->seq[pm.first + j].idx = 0;
            }
            if (ok)
                ok &= hipMemcpy(b->d_seq.p, b->seq.data(), (size_t)g * sizeof(FaSeq), hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    if (!ok) {
        set_err("falcon_amd: staging the batch failed: %s", hipGetErrorString(hipGetLastError()));
        delete b;
        return nullptr;
    }
    b->stats.L = b->sum_len;
    b->stats.T = b->sum_seed;
    b->stats.n_piles = n_pile;
    b->stats.n_seqs = g;
    return b;
}

extern "C" fa_batch *fa_batch_create(fa_ctx *ctx, int n_pile, const int *pile_n_seq,
                                     const char *const *seqs, const int *seq_len) {
    return batch_build(ctx, n_pile, pile_n_seq, seqs, seq_len, false, FA_BAND);
}

extern "C" void fa_batch_free(fa_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    {   // still waiting for the planner: its MSA stage is never begun; the planner at it: wait
        std::unique_lock<std::mutex> hold(b->ctx->plan_mu);
        auto &q = b->ctx->plan_q;
        auto it = std::find(q.begin(), q.end(), b);
        if (it != q.end()) {
            q.erase(it);
            b->back_state = 0;
        }
        b->ctx->done_cv.wait(hold, [&] { return b->back_state.load(std::memory_order_acquire) != 1; });
    }
    // Its buffers go to the per-device block cache, from where another thread's batch_build
    // may take them at once (hipFree used to wait for the device; the cache does not):
    // nothing of this batch may still be running.  Front stages (also of a submit that
    // failed half-way: k_index / k_chain / k_align / k_tags / k_links and the async copies
    // are only queued) run on the context's front stream, the back stage signals ev[6].
    // (no front_mu here: callers may hold it, and waiting for a little more than this
    // batch's own work -- whatever another thread queued since -- is harmless)
    if (b->front_launched) (void)hipStreamSynchronize(b->ctx->stream);
    if (b->back_state.load() == 2 && b->ev[6]) (void)hipEventSynchronize(b->ev[6]);
    if (b->back_state.load() == -1)  // (a stage that failed half-way may have queued part of its work)
        for (hipStream_t sb : b->ctx->back_stream)
            if (sb) (void)hipStreamSynchronize(sb);
    // (its device and pinned buffers release themselves)
    delete b;
}

// Size the alignment arena: one slot per resident wavefront.
// Cells per slot: the worst case is rows x (band + 1) (every row as wide as the band
// allows, DW_banded.c:184); alignments that end up accepted use a fraction of it (rows up
// to 0.3 (q + t) are reserved, ~0.2 are walked; the band is ~26 diagonals wide on average
// at 13 % error).  Slots are sized for FA_SLOT_WIDTH diagonals per row; an
// alignment that outgrows its slot says so (FaAln.err == 2), the launch is repeated with
// worst-case slots and the context keeps those.  The memory matters beyond its size:
// amdgpu wipes released VRAM, and the next process on the device waits for that (a 60 GB
// arena cost the next worker ~4 s).
static const int FA_SLOT_WIDTH = 32;

// hipMalloc for the context-owned arenas: what is missing may sit in the device's block
// cache (freed batches), so a failure is retried once after dropping the cache
static hipError_t arena_malloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cache_of_current_device()->drop_all();
        e = hipMalloc(p, bytes);
    }
    return e;
}

static int ensure_arena(fa_ctx *c, const fa_batch *b, size_t lds_bytes, bool full) {
    int per_cu = fa_align_blocks_per_cu(lds_bytes);
    per_cu = std::max(1, std::min(per_cu, 32));
    int n_slot = c->n_cu * per_cu;
    n_slot = std::max(1, std::min(n_slot, b->n_seq));
    if (const char *e = getenv("FALCON_AMD_SLOTS")) n_slot = std::max(1, std::min(n_slot, atoi(e)));
    u64 rows = (u64)b->max_rows;
    int width = FA_SLOT_WIDTH;
    if (const char *e = getenv("FALCON_AMD_SLOT_WIDTH")) width = std::max(1, atoi(e));  // (tests)
    if (full || width > b->band + 1) width = b->band + 1;
    u64 cells = std::max<u64>(rows * (u64)width, 4096);
    // slot stride = 4 KB x m + 256 B x 7: the slots' first pages (all waves start writing
    // at their slot's base) spread over the memory channels instead of piling on a few
    cells = ((cells + 1023) & ~(u64)1023) + 448;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // never take more than half of what is free for the transient trace arena
    u64 per_slot = cells * 4 + rows * (sizeof(FaRowRec) + sizeof(FaRowExt));
    u64 have = (u64)c->arena_cells_bytes + 2 * (u64)c->arena_rows_bytes;
    u64 budget = (u64)free_b / 2 + have;
    if ((u64)n_slot * per_slot > budget) n_slot = (int)std::max<u64>(1, budget / per_slot);
    size_t need_cells = (size_t)n_slot * cells * 4, need_rows = (size_t)n_slot * rows * sizeof(FaRowRec);
    // The slot size follows the batch's longest read, so it creeps up from batch to batch
    // of a stream; re-allocating 13 GB for a few per cent more costs 0.4-0.8 s each time
    // (and every other hipMalloc of the process queues behind it).  What is there is used
    // with fewer slots as long as that loses less than an eighth of them; a real growth
    // takes a quarter more than asked for.
    if (need_cells > c->arena_cells_bytes || need_rows > c->arena_rows_bytes) {
        const u64 fit = std::min<u64>(c->arena_cells_bytes / (cells * 4),
                                      c->arena_rows_bytes / (rows * sizeof(FaRowRec)));
        if (fit >= (u64)n_slot - (u64)n_slot / 8 && fit >= 1) {
            n_slot = (int)fit;
            need_cells = need_rows = 0;
        } else if (((u64)need_cells + 2 * (u64)need_rows) * 5 / 4 <= budget) {
            need_cells += need_cells / 4;  // (headroom only while it stays inside the budget)
            need_rows += need_rows / 4;
        }
    }
    if (need_cells > c->arena_cells_bytes || need_rows > c->arena_rows_bytes) {
        // (a pipeline drain, said out loud: submit no longer waits for the device, so the batch
        // before this one may still have its alignment kernel queued on the arena that is about
        // to be freed -- hipFree's implicit device-wide wait is not a contract to lean on)
        (void)hipStreamSynchronize(c->stream);
    }
    if (need_cells > c->arena_cells_bytes) {
        if (c->arena.cells) (void)hipFree(c->arena.cells);
        c->arena.cells = nullptr;
        c->arena_cells_bytes = 0;
        HIP_OK(arena_malloc((void **)&c->arena.cells, need_cells));
        c->arena_cells_bytes = need_cells;
    }
    if (need_rows > c->arena_rows_bytes) {
        if (c->arena.rows) (void)hipFree(c->arena.rows);
        if (c->arena.rowx) (void)hipFree(c->arena.rowx);
        c->arena.rows = nullptr;
        c->arena.rowx = nullptr;
        c->arena_rows_bytes = 0;
        HIP_OK(arena_malloc((void **)&c->arena.rows, need_rows));
        HIP_OK(arena_malloc((void **)&c->arena.rowx, need_rows));
        c->arena_rows_bytes = need_rows;
    }
    c->arena.cells_per_slot = cells;
    c->arena.rows_per_slot = rows;
    c->arena.n_slot = n_slot;
    return 0;
}

// The redo arena: min(n, 512) slots of the worst case rows x (band + 1) cells, for the
// alignments of a launch that outgrew the usual slots (a handful per batch at most: reads
// that pass the window filter and then align badly).  Repeating the whole launch with
// worst-case slots, as the first version did, meant a 60 GB allocation in the middle of a
// stream (0.7 s, and the next process waits for the driver to wipe it).
static int ensure_arena2(fa_ctx *c, const fa_batch *b, int n) {
    int n_slot = std::max(1, std::min(n, 512));
    const u64 rows = (u64)b->max_rows;
    u64 cells = std::max<u64>(rows * (u64)(b->band + 1), 4096);
    cells = ((cells + 1023) & ~(u64)1023) + 448;
    {   // worst-case slots are MBs each: never more of them than half of the free memory holds
        // (fewer slots only mean more trips of the persistent kernel)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const u64 per_slot = cells * 4 + rows * (sizeof(FaRowRec) + sizeof(FaRowExt));
        const u64 budget = (u64)free_b / 2 + (u64)c->arena2_cells_bytes + 2 * (u64)c->arena2_rows_bytes;
        if ((u64)n_slot * per_slot > budget) n_slot = (int)std::max<u64>(1, budget / per_slot);
    }
    const size_t need_cells = (size_t)n_slot * cells * 4, need_rows = (size_t)n_slot * rows * sizeof(FaRowRec);
    if (need_cells > c->arena2_cells_bytes || need_rows > c->arena2_rows_bytes)
        (void)hipEventSynchronize(c->ev_redo);  // (a repeat launch of another batch may still use the arena)
    if (need_cells > c->arena2_cells_bytes) {
        if (c->arena2.cells) (void)hipFree(c->arena2.cells);
        c->arena2.cells = nullptr;
        c->arena2_cells_bytes = 0;
        HIP_OK(arena_malloc((void **)&c->arena2.cells, need_cells));
        c->arena2_cells_bytes = need_cells;
    }
    if (need_rows > c->arena2_rows_bytes) {
        if (c->arena2.rows) (void)hipFree(c->arena2.rows);
        if (c->arena2.rowx) (void)hipFree(c->arena2.rowx);
        c->arena2.rows = nullptr;
        c->arena2.rowx = nullptr;
        c->arena2_rows_bytes = 0;
        HIP_OK(arena_malloc((void **)&c->arena2.rows, need_rows));
        HIP_OK(arena_malloc((void **)&c->arena2.rowx, need_rows));
        c->arena2_rows_bytes = need_rows;
    }
    c->arena2.cells_per_slot = cells;
    c->arena2.rows_per_slot = rows;
    c->arena2.n_slot = n_slot;
    c->arena2.counter = c->d_counter2;  // (its own: a repeat launch may run beside the next batch's k_align2)
    c->arena2.prof = c->arena.prof;
    return 0;
}

// k_align2 takes the falcon_sense alignments (band tolerance 150) and every other band it is
// built for: >= 64 (a band row never ends an alignment by its width inside a wavefront's
// lanes) and <= 190 (what the general kernel, which takes what k_align2 hands back, handles);
// base indices into the packed words are 32 bits.  FALCON_AMD_ALIGN1=1: the one alignment
// per wavefront kernel for everything (A/B and tests).
static bool use_align2(const fa_batch *b, int band) {
    const bool off = getenv("FALCON_AMD_ALIGN1") != nullptr;  // (read every time: tests switch it)
    return !off && band >= 64 && band + 1 <= 64 * FA_ALIGN_MAXCH - 1 && !b->a2_groups.empty();
}

// The tape arena of k_align2: one ring per resident wavefront, sized for the batch's
// longest alignment (twice its rows: a track may wait while its neighbour runs), at most
// 32768 iterations -- what does not fit is handed back by the kernel.  80 bytes per
// iteration: ~0.65 MB per wavefront for reads of 12 kb, 5 GB per device, where the 4-byte
// cells of k_align took 13 GB.
static int ensure_arena_a2(fa_ctx *c, const fa_batch *b) {
    u32 ring = fa_align2_ring_for(b->max_rows);
    if (const char *e = getenv("FALCON_AMD_RING")) ring = (u32)std::max(256, atoi(e));  // (tests: a power of two)
    ring = std::min<u32>(ring, 32768u);
    int per_cu = std::max(1, std::min(fa_align2_blocks_per_cu(), 32));
    int n_slot = c->n_cu * per_cu;
    n_slot = std::max(1, std::min(n_slot, (b->n_seq + 1) / 2));
    if (const char *e = getenv("FALCON_AMD_SLOTS")) n_slot = std::max(1, std::min(n_slot, atoi(e)));
    const u64 slot_words = fa_align2_slot_words(ring);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const u64 budget = (u64)free_b / 2 + (u64)c->a2_bytes;
    if ((u64)n_slot * slot_words * 4 > budget) n_slot = (int)std::max<u64>(1, budget / (slot_words * 4));
    size_t need = (size_t)n_slot * slot_words * 4;
    if (need > c->a2_bytes) {
        // fewer slots in what is there rather than a new allocation for a few per cent more
        // (ensure_arena's rule: re-allocating GBs in the middle of a stream costs 0.5 s)
        const u64 fit = c->a2_bytes / (slot_words * 4);
        if (fit >= (u64)n_slot - (u64)n_slot / 8 && fit >= 1) {
            n_slot = (int)fit;
        } else {
            if (need + need / 4 <= budget) need += need / 4;
            (void)hipStreamSynchronize(c->stream);  // (the batch before may still run on the arena: a drain, see ensure_arena)
            if (c->a2.mem) (void)hipFree(c->a2.mem);
            c->a2.mem = nullptr;
            c->a2_bytes = 0;
            HIP_OK(arena_malloc((void **)&c->a2.mem, need));
            c->a2_bytes = need;
            n_slot = (int)std::min<u64>((u64)n_slot + (u64)n_slot / 4, need / (slot_words * 4));
        }
    }
    c->a2.slot_words = slot_words;
    c->a2.ring = ring;
    c->a2.n_slot = n_slot;
    return 0;
}

// The alignment summaries are on the host (h_aln).  0: fine; 1: some alignment outgrew its
// work slot or was handed back by k_align2 (the caller repeats those with worst-case
// slots); < 0: error.
static int check_aln(fa_batch *b) {
    bool outgrown = false;
    for (int g = 0; g < b->n_seq; g++) {
        if (b->h_aln[g].err == 2) {
            outgrown = true;
        } else if (b->h_aln[g].err) {
            set_err("falcon_amd: alignment of sequence %d overflowed its work slot", g);
            return -2;
        }
    }
    b->have_aln = !outgrown;
    return outgrown ? 1 : 0;
}

// Alignment summaries to the host, through stream `s` (which is waited for).
static int fetch_aln(fa_batch *b, hipStream_t s) {
    if (b->h_aln.resize(b->n_seq)) return -1;
    HIP_OK(hipMemcpyAsync(b->h_aln.data(), b->d_aln.p, (size_t)b->n_seq * sizeof(FaAln),
                          hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return check_aln(b);
}

// Alignments that came back with FaAln.err == 2 -- they outgrew their work slot, or k_align2
// handed them back (a band that stayed wide, an alignment too long for the tape ring) -- are
// done again, alone, by the general one-alignment-per-wavefront kernel in worst-case slots
// (nothing was written past a slot), on stream `s`, which must be behind the first launch.
// The worst-case slots belong to the context: repeat launches of different batches (each on
// its own back stream) follow one another.  Returns check_aln's verdict after the launch.
static int redo_handed_back(fa_batch *b, double max_diff, int band, hipStream_t s) {
    fa_ctx *c = b->ctx;
    std::vector<int> redo;
    for (int g = 0; g < b->n_seq; g++)
        if (b->h_aln[g].err == 2) redo.push_back(g);
    if (band + 1 > 64 * FA_ALIGN_MAXCH - 1) {
        // (k_align_wide runs in worst-case slots from the start: not expected)
        set_err("falcon_amd: %zu alignments overflowed their work slots at band %d", redo.size(), band);
        return -1;
    }
    std::lock_guard<std::mutex> one(c->redo_mu);
    if (ensure_arena2(c, b, (int)redo.size()) || b->d_redo.alloc(redo.size())) return -1;
    HIP_OK(hipStreamWaitEvent(s, c->ev_redo, 0));
    HIP_OK(hipMemcpyAsync(b->d_redo.p, redo.data(), redo.size() * sizeof(int), hipMemcpyHostToDevice, s));
    const bool timed = b->ev[13] && b->ev[14];  // (runs of fa_align_pairs have no events)
    if (timed) HIP_OK(hipEventRecord(b->ev[13], s));  // (the repeat launch's own span: added to ms_align, see fill of fa_stats)
    fa_launch_align_list(b->dev(), c->arena2, b->max_read_len, b->max_seed_len, max_diff, band, b->d_redo.p,
                         (int)redo.size(), s);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(c->ev_redo, s));
    if (timed) {
        HIP_OK(hipEventRecord(b->ev[14], s));
        b->redo_timed = true;
    }
    b->stats.align_relaunched = (int)redo.size();
    // (the list must outlive the copy: fetch_aln synchronises the stream)
    int rc = fetch_aln(b, s);
    if (rc == 1) {
        set_err("falcon_amd: an alignment overflowed a worst-case work slot");
        return -2;
    }
    return rc;
}

// A run has two halves.  start_align: the stages after the windows are known (d_range on
// the device, h_range on the host or on its way there) up to the banded alignment, queued on
// the front stream; nothing in it waits for the device.  `band` <= 190 runs the tuned
// kernels, wider bands the general one; `force_accept_g` >= 0 names a sequence whose
// alignment is used whatever its length (the unitig's copy of itself, falcon.c:699-704).
// msa_stage: the MSA plan from the alignment summaries (waits for k_align) and the MSA
// kernels on a back stream -- run by the context's planner thread (planner_main), batch after
// batch in submit order; finish_run waits for its outcome.
static int start_align(fa_batch *b, unsigned min_cov, double max_diff, int band, int force_accept_g, bool two_per_wave);
static int msa_stage(fa_batch *b);
static int finish_run(fa_batch *b, bool grace);
static void begin_back(fa_batch *p);

// First half of a run (seed index, chaining, alignment) on the context's front stream;
// returns with everything only QUEUED -- it never waits for the device.  The batch then
// belongs to the context's planner thread, which waits for its alignment summaries, sizes the
// MSA pools and queues k_tags, k_links, k_score, k_backtrace on a back stream: they run beside
// the kernels of whatever was submitted next.
extern "C" int fa_batch_submit(fa_batch *b, unsigned min_cov, unsigned K, double min_idt) {
    if (!b || b->pair_mode) {
        set_err("falcon_amd: fa_batch_submit on an invalid batch");
        return -1;
    }
    if (b->in_flight) {
        set_err("falcon_amd: fa_batch_submit on a batch that is still running (fa_batch_wait first)");
        return -1;
    }
    if (K != FA_K) {
        set_err("falcon_amd: K=%u unsupported (the falcon_sense path hard-wires K=8, "
                "falcon_kit/mains/consensus.py:270)", K);
        return -1;
    }
    fa_ctx *c = b->ctx;
    HIP_OK(hipSetDevice(c->device));
    if (b->ensure_events()) {
        set_err("falcon_amd: hipEventCreate failed");
        return -1;
    }
    PhaseTimer pt("fa_batch_submit");
    {
        std::lock_guard<std::mutex> front(c->front_mu);  // one batch at a time on the front stream
        pt.mark("front-lock");
        hipStream_t s = c->stream;
        b->front_launched = true;
        b->fetched = b->fetched_eqv = false;
        b->have_range = b->have_aln = false;
        b->redo_timed = false;
        const double max_diff = 1.0 - min_idt;  // falcon.c:580
        // (decided once per run: the arena sized here is the one start_align launches on)
        const bool two_per_wave = use_align2(b, FA_BAND);
        if (!two_per_wave && b->n_seq > 0 && b->a2_groups.empty()) {
            // (not an error -- the answers are the same -- but four times the alignment time, and nobody asked for it)
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "falcon_amd: a pile of 2^28 packed words (4.29 G bases) or more: its batch is aligned by "
                                "k_align, not k_align2\n");
        }
        if (two_per_wave) {
            if (ensure_arena_a2(c, b)) return -1;
        } else {
            size_t lds = fa_align_lds_bytes(b->max_read_len, b->max_seed_len);
            if (ensure_arena(c, b, lds, false)) return -1;
        }
        pt.mark("arena");
        FaBatchDev d = b->dev();

        HIP_OK(hipEventRecord(b->ev[0], s));
        fa_launch_index(d, b->max_seed_len, s);
        HIP_OK(hipEventRecord(b->ev[1], s));
        trace_stage(s, "index");
        fa_launch_chain(d, b->max_bins, s);
        HIP_OK(hipEventRecord(b->ev[2], s));
        trace_stage(s, "chain");
        // s2 of every alignment (needed by the MSA plan) travels while k_align runs
        if (b->h_range.resize(b->n_seq)) return -1;
        HIP_OK(hipMemcpyAsync(b->h_range.data(), b->d_range.p, (size_t)b->n_seq * sizeof(FaRange),
                              hipMemcpyDeviceToHost, s));
        if (start_align(b, min_cov, max_diff, FA_BAND, -1, two_per_wave)) return -1;
        pt.mark("launch-front");
        // its second half is the planner's (a failure there is reported by fa_batch_wait)
        std::lock_guard<std::mutex> hold(c->plan_mu);
        b->back_state = 1;
        c->plan_q.push_back(b);
    }
    c->plan_cv.notify_one();
    return 0;
}

// The rest of a submitted run: waits for the back stream's kernels, checks every pile's
// verdict, fills the statistics.
extern "C" int fa_batch_wait(fa_batch *b) {
    if (!b || !b->in_flight) {
        set_err("falcon_amd: fa_batch_wait without fa_batch_submit");
        return -1;
    }
    HIP_OK(hipSetDevice(b->ctx->device));
    return finish_run(b, true);
}

extern "C" int fa_batch_run(fa_batch *b, unsigned min_cov, unsigned K, double min_idt) {
    int rc = fa_batch_submit(b, min_cov, K, min_idt);
    if (rc) return rc;
    return finish_run(b, false);
}

static int start_align(fa_batch *b, unsigned min_cov, double max_diff, int band, int force_accept_g,
                       bool two_per_wave) {
    fa_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    FaBatchDev d = b->dev();
    PhaseTimer pt("align launch");
    // (the experiment switches of the second half are read here, by the thread that called
    // submit: the planner thread must not race a test's setenv)
    b->run.score_mode = getenv("FALCON_AMD_SCORE1") ? 1 : 0;
    b->run.force_generic = getenv("FALCON_AMD_SCORE_GENERIC") ? 1 : 0;
    b->run.links_mode = getenv("FALCON_AMD_LINKS1") ? 1 : 0;
    b->run.min_cov = min_cov; b->run.max_diff = max_diff; b->run.band = band;
    b->run.force_accept_g = force_accept_g; b->run.two_per_wave = two_per_wave;
    if (b->h_aln.resize(b->n_seq) || b->h_a2_stats.resize(12)) return -1;
    if (two_per_wave) {
        (void)hipMemsetAsync(c->a2.stats, 0, 12 * sizeof(unsigned long long), s);
        for (const auto &gr : b->a2_groups)   // (one launch, unless the batch is cut: base indices are 32 bits)
            fa_launch_align2(d, c->a2, max_diff, band, b->order_dev() + gr.order_begin, gr.order_count, gr.word_base, s);
    } else if (band + 1 > 64 * FA_ALIGN_MAXCH - 1) {
        fa_launch_align_wide(d, c->arena, max_diff, band, s);
    } else {
        fa_launch_align_band(d, c->arena, b->max_read_len, b->max_seed_len, max_diff, band, s);
    }
    HIP_OK(hipEventRecord(b->ev[3], s));
    trace_stage(s, "align");
    if (getenv("FALCON_AMD_PROF")) {  // only meaningful in -DFA_ALIGN_PROF builds
        u64 pf[8];
        HIP_OK(hipStreamSynchronize(s));
        HIP_OK(hipMemcpy(pf, c->arena.prof, sizeof(pf), hipMemcpyDeviceToHost));
        HIP_OK(hipMemset(c->arena.prof, 0, sizeof(pf)));
        fprintf(stderr, "[falcon_amd] k_align section ticks:");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", (unsigned long long)pf[i]);
        fprintf(stderr, "\n");
    }
    HIP_OK(hipGetLastError());
    // the alignment summaries (they bound the MSA node pools) and the kernel's own counters
    // leave for the host as soon as the kernel is done (pinned buffers: true asynchronous copies)
    HIP_OK(hipMemcpyAsync(b->h_aln.data(), b->d_aln.p, (size_t)b->n_seq * sizeof(FaAln),
                          hipMemcpyDeviceToHost, s));
    if (two_per_wave)
        HIP_OK(hipMemcpyAsync(b->h_a2_stats.data(), c->a2.stats, 12 * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, s));
    HIP_OK(hipEventRecord(b->ev[12], s));
    b->stats.align_relaunched = 0;
    b->stats.align_slots = two_per_wave ? c->a2.n_slot : c->arena.n_slot;
    b->stats.align_slot_cells = two_per_wave ? (long long)c->a2.ring * 64 : (long long)c->arena.cells_per_slot;
    // ---- while k_align runs: the part of the MSA plan that only depends on the seed
    // lengths (segment work list of k_links, per-pile offsets of the per-position arrays)
    const int TSEG = 128;  // must match k_msa.hip
    u64 t_tot = 0;
    size_t n_seg = 0;
    for (int p = 0; p < b->n_pile; p++) {
        t_tot += (u64)b->pile[p].seed_len;
        n_seg += (size_t)((b->pile[p].seed_len + TSEG - 1) / TSEG);
    }
    b->run.t_tot = t_tot;
    b->run.n_seg = n_seg;
    if (!b->msa_static) {
        std::vector<int> seg_pile, seg_t0;
        std::vector<u64> t_off(b->n_pile);
        std::vector<u32> seg_first(b->n_pile + 1);
        seg_pile.reserve(n_seg); seg_t0.reserve(n_seg);
        u64 tt = 0;
        for (int p = 0; p < b->n_pile; p++) {
            t_off[p] = tt;
            seg_first[p] = (u32)seg_pile.size();
            tt += (u64)b->pile[p].seed_len;
            for (int t0 = 0; t0 < b->pile[p].seed_len; t0 += TSEG) {
                seg_pile.push_back(p);
                seg_t0.push_back(t0);
            }
        }
        if (b->d_seg_pile.alloc(n_seg + 1) || b->d_seg_t0.alloc(n_seg + 1) ||
            b->d_wide.alloc(6 * (n_seg + 1) + 1) || b->d_t_off.alloc((size_t)b->n_pile + 1) ||
            b->d_seg_cnt.alloc(2 * n_seg + 2) || b->d_seg_base.alloc(2 * n_seg + 2) ||
            b->d_seg_first.alloc((size_t)b->n_pile + 1) || b->d_bound.alloc((size_t)b->n_pile + 1))
            return -1;
        // (synchronous copies on the null stream; the context's streams are non-blocking,
        // so they do not wait for k_align)
        HIP_OK(hipMemcpy(b->d_seg_pile.p, seg_pile.data(), n_seg * sizeof(int), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b->d_seg_t0.p, seg_t0.data(), n_seg * sizeof(int), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b->d_t_off.p, t_off.data(), t_off.size() * sizeof(u64), hipMemcpyHostToDevice));
        seg_first[b->n_pile] = (u32)seg_pile.size();
        HIP_OK(hipMemcpy(b->d_seg_first.p, seg_first.data(), seg_first.size() * sizeof(u32), hipMemcpyHostToDevice));
        b->msa_static = true;
    }
    if (b->h_ta.resize((size_t)b->n_seq + 1) || b->h_acc_first.resize((size_t)b->n_pile + 1) ||
        b->h_link_off.resize((size_t)b->n_pile) || b->h_link_cap.resize((size_t)b->n_pile) ||
        b->h_pile_up.resize((size_t)b->n_pile) || b->h_pile_out.resize(b->n_pile))
        return -1;
    pt.mark("launch+static-plan");
    b->back_state = 0;
    b->back_err.clear();
    b->in_flight = true;
    return 0;
}

// Second half of a run, by the one thread the batch was handed to (back_state 1): the MSA plan from the
// alignment summaries (host, O(#reads)), then k_tags | k_tscan | k_links | k_score |
// k_backtrace and the per-pile results' copy on a back stream.  Nothing here touches the
// front stream, which meanwhile runs the next batch.
static int msa_stage(fa_batch *b) {
    fa_ctx *c = b->ctx;
    PhaseTimer pt("msa stage");
    const unsigned min_cov = b->run.min_cov;
    const int force_accept_g = b->run.force_accept_g;
    hipStream_t sb;
    {
        std::lock_guard<std::mutex> hold(c->back_mu);
        sb = c->back_stream[c->back_turn++ % (unsigned)c->n_back];
    }
    HIP_OK(hipEventSynchronize(b->ev[12]));
    pt.mark("align-wait+summaries");
    int rc_aln = check_aln(b);
    HIP_OK(hipStreamWaitEvent(sb, b->ev[3], 0));
    if (rc_aln == 1) rc_aln = redo_handed_back(b, b->run.max_diff, b->run.band, sb);
    if (rc_aln) return rc_aln;
    b->have_range = true;  // its copy was queued ahead of k_align
    if (force_accept_g >= 0 && b->h_aln[force_accept_g].aligned) b->h_aln[force_accept_g].accept = 1;
    // ---- plan the MSA stage from the alignment summaries
    u64 node_off = 0, desc_tot = 4, ins_tot = 0, link_tot = 0;  // (desc: front padding, k_links loads groups of 4)
    long long sC = 0, sD = 0, sA = 0, nal = 0;
    FaTagAln *ta = b->h_ta.data();
    size_t n_ta = 0;
    u32 *acc_first = b->h_acc_first.data();
    u64 *link_off = b->h_link_off.data(), *link_cap = b->h_link_cap.data();
    for (int p = 0; p < b->n_pile; p++) {
        FaPile &pm = b->pile[p];
        u64 levels = (u64)pm.seed_len + 2, cols = 8;
        acc_first[p] = (u32)n_ta;
        int n_acc = 0;
        b->pile_err[p] = b->pile_fixed_err[p];
        // what the pile adds to the pools, committed below unless the pile is too deep
        const size_t n_ta0 = n_ta;
        const u64 desc0 = desc_tot, ins0 = ins_tot;
        const long long sD0 = sD, sA0 = sA;
        for (int j = 1; j < pm.n_seq; j++) {
            const int g = pm.first + j;
            const FaAln &al = b->h_aln[g];
            sC += al.cells;
            if (!al.accept) continue;
            levels += (u64)al.n_ins;
            cols += (u64)al.size;
            sD += al.dist;
            sA += al.size;
            n_acc++;
            FaTagAln &x = ta[n_ta++];
            x.desc_off = desc_tot + 1;  // (the slot before: a leading insertion run, k_tags)
            x.ins_off = (u32)ins_tot;
            x.s2 = b->h_range[g].s2;  // start of the alignment on the seed (chain stage)
            x.g = g;
            x.pile = p;
            x.pad = 0;
            desc_tot += (u64)al.t_e + 3;
            ins_tot += (u64)al.n_ins + 4;
        }
        if (n_acc > FA_CNS_MAX_ALN) {
            // deeper than the MSA kernels handle: this pile alone is reported
            // (fa_batch_pile_error) and gets no consensus; the batch goes on without it
            b->pile_err[p] = 2;
            b->pile_err_arg[p] = n_acc;
            n_ta = n_ta0; desc_tot = desc0; ins_tot = ins0; sD = sD0; sA = sA0;
            n_acc = 0;
            levels = (u64)pm.seed_len + 2;
            cols = 8;
        }
        nal += n_acc;
        pm.node_off = node_off;
        pm.node_cap = levels * 5;
        node_off += pm.node_cap;
        link_off[p] = link_tot;
        link_cap[p] = cols;
        link_tot += cols;
        b->h_pile_up[p] = pm;
    }
    acc_first[b->n_pile] = (u32)n_ta;
    b->n_ta = n_ta;
    pt.mark("plan");
    if (ins_tot >= 0xffffffffull || link_tot >= 0xffffffffull * 4) {
        set_err("falcon_amd: batch too large for the MSA stage");
        return -1;
    }
    const u64 t_tot = b->run.t_tot;
    const size_t n_seg = b->run.n_seg;
    auto need = [&](auto &buf, size_t n) { return (buf.n < n) ? buf.alloc(n) : 0; };
    int rc2 = 0;
    rc2 |= need(b->d_ta, n_ta + 1);
    rc2 |= need(b->d_acc_first, (size_t)b->n_pile + 1);
    rc2 |= need(b->d_tcov, n_ta + 1);
    rc2 |= need(b->d_desc, (size_t)desc_tot + 8);
    rc2 |= need(b->d_insb, (size_t)ins_tot + 8);
    rc2 |= need(b->d_tinfo, (size_t)t_tot + 8);
    rc2 |= need(b->d_links, (size_t)link_tot + 8);
    rc2 |= need(b->d_link_off, (size_t)b->n_pile);
    rc2 |= need(b->d_link_cap, (size_t)b->n_pile);
    rc2 |= need(b->d_lvl_nlink, (size_t)(node_off / 5) + 8);
    rc2 |= need(b->d_score_ovf, (size_t)b->n_pile * 2 * 256 * 5);
    rc2 |= need(b->d_score_out, (size_t)b->n_pile);
    rc2 |= need(b->d_nodes, (size_t)node_off + 8);
    if (rc2) return -1;
    pt.mark("buffers");
    // (pinned sources: the copies are queued, not staged -- a staged copy would wait for
    // whatever the back stream still runs of the batch two before this one)
    auto up = [&](void *dst, const void *src, size_t bytes) {
        return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sb) == hipSuccess;
    };
    bool okc = true;
    okc &= up(b->d_ta.p, ta, n_ta * sizeof(FaTagAln));
    okc &= up(b->d_acc_first.p, acc_first, ((size_t)b->n_pile + 1) * sizeof(u32));
    okc &= up(b->d_link_off.p, link_off, (size_t)b->n_pile * sizeof(u64));
    okc &= up(b->d_link_cap.p, link_cap, (size_t)b->n_pile * sizeof(u64));
    okc &= up(b->d_pile.p, b->h_pile_up.data(), (size_t)b->n_pile * sizeof(FaPile));
    if (!okc) {
        set_err("falcon_amd: uploading the MSA plan failed");
        return -1;
    }
    FaMsaDev md;
    md.ta = b->d_ta.p; md.acc_first = b->d_acc_first.p; md.n_acc_total = (int)n_ta;
    md.tcov = b->d_tcov.p; md.desc = b->d_desc.p; md.insb = b->d_insb.p; md.seg_cnt = b->d_seg_cnt.p;
    md.seg_base = b->d_seg_base.p; md.seg_first = b->d_seg_first.p; md.bound = b->d_bound.p; md.t_off = b->d_t_off.p; md.tinfo = b->d_tinfo.p;
    md.links = b->d_links.p; md.link_off = b->d_link_off.p; md.link_cap = b->d_link_cap.p;
    md.lvl_nlink16 = b->d_lvl_nlink.p; md.score_ovf = b->d_score_ovf.p;
    md.score_out = b->d_score_out.p; md.seg_pile = b->d_seg_pile.p; md.seg_t0 = b->d_seg_t0.p;
    md.n_seg = (int)n_seg;
    md.wide_count = b->d_wide.p; md.wide_list = b->d_wide.p + 1;
    md.first_links_back = force_accept_g >= 0 ? 1 : 0;
    md.force_generic = b->run.force_generic;  // (tests: pins the generic path of k_score1)
    md.score_mode = b->run.score_mode;        // (FALCON_AMD_SCORE1: k_score1 for every pile)
    md.links_mode = b->run.links_mode;        // (FALCON_AMD_LINKS1: k_links for every segment)
    const FaBatchDev d = b->dev();
    pt.mark("upload");
    HIP_OK(hipEventRecord(b->ev[4], sb));
    fa_launch_msa_front(d, md, min_cov, sb, b->ev[8], b->ev[9]);  // k_tags + k_tscan | k_links
    HIP_OK(hipEventRecord(b->ev[5], sb));
    trace_stage(sb, "links");
    HIP_OK(hipEventRecord(b->ev[7], sb));
    fa_launch_msa_back(d, md, min_cov, sb, b->ev[10], b->ev[11]);  // k_score | k_backtrace
    trace_stage(sb, "consensus");
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(b->h_pile_out.data(), b->d_pile_out.p,
                          (size_t)b->n_pile * sizeof(FaPileOut), hipMemcpyDeviceToHost, sb));
    HIP_OK(hipEventRecord(b->ev[6], sb));
    pt.mark("msa-launch");
    // what the statistics need of this plan
    b->stats.C = sC; b->stats.D = sD; b->stats.A = sA; b->stats.n_aligned = nal;
    if (b->run.two_per_wave) {
        const unsigned long long *st = b->h_a2_stats.data();  // (landed with the summaries)
        b->stats.align_arena_bytes = (long long)c->a2_bytes + (long long)c->arena2_cells_bytes +
                                     2 * (long long)c->arena2_rows_bytes;
        b->stats.align_pair_iterations = (long long)st[0];
        b->stats.align_single_iterations = (long long)st[1];
        b->stats.align_placements = (long long)st[2];
        b->stats.align_parkings = (long long)st[3];
        b->stats.align_handed_back = (long long)st[4];
        b->stats.align_wide_rows = (long long)st[5];
        b->stats.align_replacements = (long long)st[7];
        b->stats.align_handed_back_tape = (long long)st[8];
        b->stats.align_handed_back_wide = (long long)st[9];
        b->stats.align_handed_back_escapes = (long long)st[10];
    } else {
        b->stats.align_arena_bytes = (long long)c->arena_cells_bytes + 2 * (long long)c->arena_rows_bytes +
                                     (long long)c->arena2_cells_bytes + 2 * (long long)c->arena2_rows_bytes;
        b->stats.align_pair_iterations = b->stats.align_single_iterations = b->stats.align_placements = 0;
        b->stats.align_parkings = b->stats.align_handed_back = b->stats.align_wide_rows = 0;
        b->stats.align_replacements = 0;
        b->stats.align_handed_back_tape = b->stats.align_handed_back_wide = b->stats.align_handed_back_escapes = 0;
    }
    return 0;
}

// Run the second half of `p` (back_state 1) and publish the outcome; a failure is kept with
// the batch, for the thread that waits for it.
static void begin_back(fa_batch *p) {
    const std::string mine = g_err;  // (another batch's failure is not this call's)
    const int rc = msa_stage(p);
    if (rc) {
        p->back_err = g_err;
        p->back_rc = rc;
        g_err = mine;
    }
    // (once the state is published the batch is its waiter's again, who may free it at once -- a
    // spurious or another batch's wake-up is enough: nothing of `p` is touched after the store.
    // Found by the ThreadSanitizer build of this file, tests/san/engine_san.cpp.)
    fa_ctx *c = p->ctx;
    {
        std::lock_guard<std::mutex> hold(c->plan_mu);
        p->back_state.store(rc ? -1 : 2, std::memory_order_release);
    }
    c->done_cv.notify_all();
}

// The planner thread of a context: the second halves of the submitted runs, in submit order.
static void planner_main(fa_ctx *c) {
    (void)hipSetDevice(c->device);
    for (;;) {
        fa_batch *p = nullptr;
        {
            std::unique_lock<std::mutex> hold(c->plan_mu);
            c->plan_cv.wait(hold, [&] { return c->plan_stop || !c->plan_q.empty(); });
            if (c->plan_q.empty()) return;  // (stop, and nothing left to do)
            p = c->plan_q.front();
            c->plan_q.pop_front();
        }
        begin_back(p);
    }
}

static int finish_run(fa_batch *b, bool grace) {
    // (the second half is the planner's, or -- unitig runs -- was the caller's own)
    (void)grace;
    fa_ctx *c = b->ctx;
    {
        std::unique_lock<std::mutex> hold(c->plan_mu);
        if (b->back_state.load(std::memory_order_acquire) == 0) {
            b->in_flight = false;
            set_err("falcon_amd: the batch's consensus stage was never queued");
            return -1;
        }
        c->done_cv.wait(hold, [&] {
            const int st = b->back_state.load(std::memory_order_acquire);
            return st == 2 || st == -1;
        });
    }
    b->in_flight = false;
    if (b->back_state.load() == -1) {
        g_err = b->back_err;
        return b->back_rc ? b->back_rc : -1;
    }
    HIP_OK(hipEventSynchronize(b->ev[6]));
    long long sO = 0;
    int n_failed = 0;
    for (int p = 0; p < b->n_pile; p++) {
        if (b->h_pile_out[p].err && !b->pile_err[p]) {  // (the MSA pools are sized from exact
            b->pile_err[p] = 1;                         // bounds: not expected, but contained)
            b->pile_err_arg[p] = b->h_pile_out[p].err;
        }
        if (b->pile_err[p]) {
            b->h_pile_out[p].len = 0;
            n_failed++;
            continue;
        }
        sO += b->h_pile_out[p].len;
    }
    b->stats.n_piles_failed = n_failed;
    fa_stats &st = b->stats;
    st.O = sO;
    (void)hipEventElapsedTime(&st.ms_index, b->ev[0], b->ev[1]);
    (void)hipEventElapsedTime(&st.ms_chain, b->ev[1], b->ev[2]);
    (void)hipEventElapsedTime(&st.ms_align, b->ev[2], b->ev[3]);
    if (b->redo_timed) {  // the alignments k_align2 handed back, repeated on the back stream
        float ms_redo = 0;
        if (hipEventElapsedTime(&ms_redo, b->ev[13], b->ev[14]) == hipSuccess) st.ms_align += ms_redo;
    }
    (void)hipEventElapsedTime(&st.ms_tags, b->ev[4], b->ev[8]);
    (void)hipEventElapsedTime(&st.ms_links, b->ev[8], b->ev[9]);
    (void)hipEventElapsedTime(&st.ms_score, b->ev[7], b->ev[10]);
    (void)hipEventElapsedTime(&st.ms_backtrace, b->ev[10], b->ev[11]);
    // the consensus stage = its four kernels (they may have run beside another batch's front
    // stages); total = first launch to last result, whatever ran in between
    st.ms_consensus = st.ms_tags + st.ms_links + st.ms_score + st.ms_backtrace;
    (void)hipEventElapsedTime(&st.ms_total, b->ev[0], b->ev[6]);
    return 0;
}

// Unitig consensus (falcon.c:668-773 generate_utg_consensus): reads are placed on the
// unitig by the caller's offsets instead of k-mer chaining, aligned with band 500, and
// the unitig itself takes part as an all-match alignment; min_cov is 0 (:755).
extern "C" fa_batch *fa_utg_consensus(fa_ctx *ctx, int n_seq, const char *const *seqs, int *offset,
                                      double min_idt) {
    if (!ctx || n_seq < 1 || !seqs) {
        set_err("falcon_amd: fa_utg_consensus: bad arguments");
        return nullptr;
    }
    // pile = [unitig, unitig again (its identity alignment, :699-704), reads 1..n_seq-1]
    const int n = n_seq + 1;
    std::vector<const char *> sq((size_t)n);
    std::vector<int> ln((size_t)n);
    sq[0] = sq[1] = seqs[0];
    ln[0] = ln[1] = (int)strlen(seqs[0]);
    for (int j = 1; j < n_seq; j++) {
        sq[j + 1] = seqs[j];
        ln[j + 1] = (int)strlen(seqs[j]);
    }
    const int band = 500;  // :723-745
    fa_batch *b = batch_build(ctx, 1, &n, sq.data(), ln.data(), false, band);
    if (!b) return nullptr;
    auto fail = [&](const char *what) -> fa_batch * {
        if (what) set_err("falcon_amd: fa_utg_consensus: %s", what);
        fa_batch_free(b);
        return nullptr;
    };
    const int utg_len = ln[0];
    if (b->h_range.resize(b->n_seq)) return fail(nullptr);
    FaRange *rg = b->h_range.data();
    memset(rg, 0, (size_t)b->n_seq * sizeof(FaRange));
    rg[1].e1 = rg[1].e2 = utg_len;
    rg[1].ok = 1;
    for (int j = 1; j < n_seq; j++) {
        FaRange &r = rg[j + 1];
        const int r_len = ln[j + 1];
        int len;
        if (offset[j] < 0) {  // the read starts before the unitig (:712-731)
            if (r_len + offset[j] < 128) continue;
            len = (r_len + offset[j] < utg_len) ? r_len + offset[j] : utg_len;
            r.s1 = -offset[j];
            r.s2 = 0;
            offset[j] = 0;  // the reference rewrites the caller's array too (:731)
        } else {
            if (offset[j] > utg_len - 128) continue;
            len = (offset[j] + r_len > utg_len) ? utg_len - offset[j] : r_len;
            r.s1 = 0;
            r.s2 = offset[j];
        }
        if (r.s1 + len > r_len) len = r_len - r.s1;  // (the reference would read past the read's end)
        if (len <= 0) continue;
        r.e1 = r.s1 + len;
        r.e2 = r.s2 + len;
        r.ok = 1;
    }
    fa_ctx *c = ctx;
    if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice failed");
    hipStream_t s = c->stream;
    b->front_launched = true;
    if (hipMemcpyAsync(b->d_range.p, rg, (size_t)b->n_seq * sizeof(FaRange), hipMemcpyHostToDevice, s) !=
        hipSuccess)
        return fail("range upload failed");
    b->fetched = b->fetched_eqv = false;
    b->have_aln = false;
    if (b->ensure_events()) return fail("hipEventCreate failed");
    {
        std::lock_guard<std::mutex> front(c->front_mu);
        b->front_launched = true;
        // (batches submitted on this context and not yet waited for are the planner's; this
        // run's second half is done by the calling thread, below)
        if (ensure_arena(c, b, fa_align_lds_bytes(b->max_read_len, b->max_seed_len), true)) return fail(nullptr);
        for (int i = 0; i < 3; i++) (void)hipEventRecord(b->ev[i], s);
        if (start_align(b, 0, 1.0 - min_idt, band, 1, false)) return fail(nullptr);
        b->back_state = 1;
    }
    begin_back(b);
    if (finish_run(b, false)) return fail(nullptr);
    return b;
}

// --trim windows of every read (consensus.py:48-99 get_alignment): seed index, then
// k_trimwin twice -- a counting pass sizes the per-wavefront scratch of the real one.
extern "C" int fa_batch_trim_windows(fa_batch *b, unsigned K, int mask_threshold) {
    if (!b || b->pair_mode) {
        set_err("falcon_amd: fa_batch_trim_windows on an invalid batch");
        return -1;
    }
    if (K != FA_K) {
        set_err("falcon_amd: K=%u unsupported (the falcon_sense path hard-wires K=8, "
                "falcon_kit/mains/consensus.py:50)", K);
        return -1;
    }
    fa_ctx *c = b->ctx;
    HIP_OK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> front(c->front_mu);
    hipStream_t s = c->stream;
    b->have_range = false;
    FaBatchDev d = b->dev();
    DevBuf<int> counter;
    if (counter.alloc(4)) return -1;
    const int n_slot = std::max(1, c->n_cu) * 4;  // 32 KB of LDS per wavefront
    fa_launch_index(d, b->max_seed_len, s);
    fa_launch_trimwin(d, n_slot, counter.p, nullptr, 0, mask_threshold, 1, s);
    if (b->h_range.resize(b->n_seq)) return -1;
    HIP_OK(hipMemcpyAsync(b->h_range.data(), b->d_range.p, (size_t)b->n_seq * sizeof(FaRange),
                          hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    long long n_max = 0;
    for (int g = 0; g < b->n_seq; g++) n_max = std::max<long long>(n_max, b->h_range[g].n_hit);
    u64 cap = 0;
    DevBuf<u32> scratch;
    if (n_max > 2048) {  // TW_LDS_N: larger hit lists work in an HBM slot per wavefront
        cap = 4096;
        while ((long long)cap < n_max) cap <<= 1;
        if (scratch.alloc((size_t)n_slot * 4 * cap)) return -1;
    }
    fa_launch_trimwin(d, n_slot, counter.p, scratch.p, cap, mask_threshold, 0, s);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(b->h_range.data(), b->d_range.p, (size_t)b->n_seq * sizeof(FaRange),
                          hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    counter.release();
    scratch.release();
    for (int g = 0; g < b->n_seq; g++) {
        if (b->h_range[g].ok < 0) {
            set_err("falcon_amd: trim window of sequence %d overflowed its scratch", g);
            return -2;
        }
    }
    b->have_range = true;
    return 0;
}

extern "C" int fa_batch_fetch(fa_batch *b, int want_eqv) {
    if (!b || b->pair_mode || b->h_pile_out.empty() || b->in_flight) {
        set_err("falcon_amd: fa_batch_fetch before a completed fa_batch_run");
        return -1;
    }
    fa_ctx *c = b->ctx;
    HIP_OK(hipSetDevice(c->device));
    // Through the context's pinned landing buffer, on the download stream: a true DMA
    // transfer that queues behind no kernel.  A pile's consensus sits right-aligned in its
    // 2T slots; what is used of a batch is one contiguous span per pile, fetched whole
    // (the slots are contiguous) and cut up on the host.
    std::lock_guard<std::mutex> hold(c->fetch_mu);
    const size_t per_slot = want_eqv ? 1 + sizeof(int) : 1;
    const size_t need = (b->out_slots + 8) * per_slot;
    if (need > c->h_dl_cap) {
        if (c->h_dl) (void)hipHostFree(c->h_dl);
        c->h_dl = nullptr;
        c->h_dl_cap = 0;
        const size_t cap = need + need / 8;
        HIP_OK(hipHostMalloc((void **)&c->h_dl, cap, hipHostMallocDefault));
        c->h_dl_cap = cap;
    }
    char *h_seq = c->h_dl;
    int *h_eqv = reinterpret_cast<int *>(c->h_dl + ((b->out_slots + 8 + 15) & ~(size_t)15));
    HIP_OK(hipMemcpyAsync(h_seq, b->d_out_seq.p, b->out_slots, hipMemcpyDeviceToHost, c->dl_stream));
    if (want_eqv)
        HIP_OK(hipMemcpyAsync(h_eqv, b->d_out_eqv.p, b->out_slots * sizeof(int), hipMemcpyDeviceToHost,
                              c->dl_stream));
    HIP_OK(hipStreamSynchronize(c->dl_stream));
    b->h_result.assign(b->n_pile, std::string());
    for (int p = 0; p < b->n_pile; p++) {
        const FaPileOut &po = b->h_pile_out[p];
        b->h_result[p].assign(h_seq + b->pile[p].out_off + po.start, (size_t)po.len);
    }
    if (want_eqv) b->h_out_eqv.assign(h_eqv, h_eqv + b->out_slots);
    b->fetched = true;
    b->fetched_eqv = want_eqv != 0;
    return 0;
}

extern "C" int fa_batch_result(fa_batch *b, int p, const char **seq, int *len, const int **eqv) {
    if (!b || !b->fetched || p < 0 || p >= b->n_pile) {
        set_err("falcon_amd: fa_batch_result: no fetched result for pile %d", p);
        return -1;
    }
    if (seq) *seq = b->h_result[p].c_str();
    if (len) *len = (int)b->h_result[p].size();
    if (eqv) {
        if (!b->fetched_eqv) {
            set_err("falcon_amd: eqv was not fetched");
            return -1;
        }
        *eqv = b->h_out_eqv.data() + b->pile[p].out_off + b->h_pile_out[p].start;
    }
    return 0;
}

extern "C" int fa_batch_fasta(fa_batch *b, const char *const *seed_ids, int mode, const char **text,
                              long long *len) {
    if (!b || !b->fetched || !seed_ids || mode < 0 || mode > 2) {
        set_err("falcon_amd: fa_batch_fasta: no fetched batch, or bad arguments");
        return -1;
    }
    std::string &out = b->fasta;
    out.clear();
    size_t total = 0;
    for (int p = 0; p < b->n_pile; p++) total += b->h_result[p].size();
    out.reserve(total + total / 64 + 64 * (size_t)b->n_pile);
    for (int p = 0; p < b->n_pile; p++)
        if (!b->pile_err[p])
            fa_fasta_append(out, seed_ids[p], b->h_result[p].data(), (long long)b->h_result[p].size(), mode);
    if (text) *text = out.data();
    if (len) *len = (long long)out.size();
    return 0;
}

extern "C" int fa_batch_pile_error(fa_batch *b, int p, char *msg, int msg_cap) {
    if (!b || p < 0 || p >= b->n_pile) return -1;
    const int code = b->pile_err[p];
    if (code && msg && msg_cap > 0) {
        if (code == 3)
            snprintf(msg, (size_t)msg_cap, "%s", b->pile_err_msg.count(p) ? b->pile_err_msg[p].c_str() : "a byte other than ACGT");
        else if (code == 2)
            snprintf(msg, (size_t)msg_cap, "%d usable reads; the GPU consensus stage handles at most %d per "
                     "pile (the reference has no such limit: lower --max-n-read)", b->pile_err_arg[p],
                     FA_CNS_MAX_ALN);
        else
            snprintf(msg, (size_t)msg_cap, "consensus stage failed (device code %d: MSA pool overflow)",
                     b->pile_err_arg[p]);
    }
    return code;
}

extern "C" int fa_batch_stats(fa_batch *b, fa_stats *out) {
    if (!b || !out) return -1;
    *out = b->stats;
    return 0;
}

extern "C" int fa_batch_range(fa_batch *b, int g, int *s1, int *e1, int *s2, int *e2,
                              long long *score, int *ok, int *n_hit) {
    if (!b || g < 0 || g >= b->n_seq) return -1;
    if (!b->have_range) {
        b->h_range.resize(b->n_seq);
        HIP_OK(hipSetDevice(b->ctx->device));
        HIP_OK(hipMemcpy(b->h_range.data(), b->d_range.p, (size_t)b->n_seq * sizeof(FaRange),
                         hipMemcpyDeviceToHost));
        b->have_range = true;
    }
    const FaRange &r = b->h_range[g];
    if (s1) *s1 = r.s1;
    if (e1) *e1 = r.e1;
    if (s2) *s2 = r.s2;
    if (e2) *e2 = r.e2;
    if (score) *score = r.score;
    if (ok) *ok = r.ok;
    if (n_hit) *n_hit = r.n_hit;
    return 0;
}

extern "C" int fa_batch_alignment(fa_batch *b, int g, int *dist, int *q_e, int *t_e, int *size,
                                  int *accept, long long *cells) {
    if (!b || g < 0 || g >= b->n_seq || !b->have_aln) return -1;
    const FaAln &a = b->h_aln[g];
    if (dist) *dist = a.dist;
    if (q_e) *q_e = a.q_e;
    if (t_e) *t_e = a.t_e;
    if (size) *size = a.size;
    if (accept) *accept = a.accept;
    if (cells) *cells = a.cells;
    return 0;
}

// Diagnostics: the host packer on one sequence (tests, no device needed).
extern "C" int fa_debug_pack(const char *s, int len, unsigned *out, long long n_out) {
    if (!s || len < 0 || !out || n_out < (long long)((len + 15) / 16)) return -2;
    return fa_pack_host(s, len, out, n_out);
}

// Diagnostics: the k-mer hits of sequence g against its pile's seed, as the chain stage
// enumerates them (find_kmer_pos_for_seq, kmer_lookup.c:207-286: query offsets 0, 4, 8, ..
// ascending, seed positions of a k-mer ascending) -- rebuilt on the host from exactly what
// k_chain's later passes walk: the per-probe bucket bounds its first pass left in the probe
// buffer, and the seed index's position lists.  Returns the number of hits (which may be
// more than `cap`, the number stored), or -1.
extern "C" int fa_batch_debug_hits(fa_batch *b, int g, int *q_pos, int *t_pos, int cap) {
    if (!b || b->pair_mode || g < 0 || g >= b->n_seq || b->in_flight || !b->have_range) {
        set_err("falcon_amd: fa_batch_debug_hits needs a completed run and a valid sequence index");
        return -1;
    }
    HIP_OK(hipSetDevice(b->ctx->device));
    const FaSeq &sq = b->seq[g];
    if (sq.idx == 0) return 0;  // the seed itself
    const FaPile &pm = b->pile[sq.pile];
    const int n_probe = (sq.len > FA_K) ? (sq.len - FA_K + 3) / 4 : 0;
    if (n_probe == 0) return 0;
    std::vector<u64> pr((size_t)n_probe);
    std::vector<u32> pos((size_t)pm.seed_len + 4);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(pr.data(), b->d_probe.p + b->probe_off[g], pr.size() * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pos.data(), b->d_kpos.p + pm.kpos_off, pos.size() * sizeof(u32), hipMemcpyDeviceToHost));
    long long n = 0;
    for (int p = 0; p < n_probe; p++) {
        // k_chain.hip's record: n = min(hits, 3); n <= 2: the hits themselves, n = 3: the bucket
        const u32 nn = (u32)pr[p] & 3u, fa = (u32)(pr[p] >> 2) & 0x1ffffu, fb = (u32)(pr[p] >> 19) & 0x1ffffu;
        const u32 cnt = nn == 3 ? fb : nn;
        for (u32 e = 0; e < cnt; e++, n++) {
            if (n < cap) {
                if (nn == 3 && fa + e >= pos.size()) {
                    set_err("falcon_amd: probe %d of sequence %d points outside its pile's index", p, g);
                    return -1;
                }
                q_pos[n] = 4 * p;
                t_pos[n] = (int)(nn == 3 ? pos[fa + e] : e == 0 ? fa : fb);
            }
        }
    }
    return (int)std::min<long long>(n, 0x7fffffff);
}

// Diagnostics: the tag words k_tags made of the accepted alignment of sequence g (one per
// covered seed position, k_msa.hip: bit 31 the read deletes the seed base, bits 30..23 the
// length of the insertion run behind it, bits 22..0 the run -- up to 11 bases inline, 2 bits
// each, first base lowest; a longer run: the index of its first base in `ins`), the word of
// the slot before the alignment's first position (a leading insertion run; 0: none) and the
// alignment's inserted bases (codes 0..3).  Returns the number of covered positions (words
// stored: min(that, cap_words)), 0 for an alignment that was not accepted, -1 on error.
extern "C" int fa_batch_debug_tags(fa_batch *b, int g, unsigned *words, int cap_words, unsigned *lead_word,
                                   unsigned char *ins, int cap_ins, int *n_ins) {
    if (!b || b->pair_mode || g < 0 || g >= b->n_seq || b->in_flight || !b->have_aln ||
        b->back_state.load() != 2) {
        set_err("falcon_amd: fa_batch_debug_tags needs a completed run and a valid sequence index");
        return -1;
    }
    HIP_OK(hipSetDevice(b->ctx->device));
    const FaTagAln *ta = b->h_ta.data();
    const FaTagAln *mine = nullptr;
    int k = 0;
    for (size_t i = 0; i < b->n_ta; i++)
        if (ta[i].g == g) { mine = &ta[i]; k = (int)i; }
    if (!mine) return 0;
    const FaAln &al = b->h_aln[g];
    HIP_OK(hipDeviceSynchronize());
    int tcov = 0;
    HIP_OK(hipMemcpy(&tcov, b->d_tcov.p + k, sizeof(int), hipMemcpyDeviceToHost));
    const int n_cov = tcov & 0x3fffffff;  // (bit 30: the alignment opens with an insertion run)
    if (n_cov > al.t_e + 2) {
        set_err("falcon_amd: alignment %d covers %d positions but ends at %d", g, n_cov, al.t_e);
        return -1;
    }
    if (lead_word) HIP_OK(hipMemcpy(lead_word, b->d_desc.p + mine->desc_off - 1, sizeof(u32), hipMemcpyDeviceToHost));
    const int nw = std::min(n_cov, cap_words);
    if (nw > 0) HIP_OK(hipMemcpy(words, b->d_desc.p + mine->desc_off, (size_t)nw * sizeof(u32), hipMemcpyDeviceToHost));
    if (n_ins) *n_ins = al.n_ins;
    const int ni = std::min(al.n_ins, cap_ins);
    if (ni > 0) HIP_OK(hipMemcpy(ins, b->d_insb.p + mine->ins_off, (size_t)ni, hipMemcpyDeviceToHost));
    return n_cov;
}

// ---------------------------------------------------------------------------
// Pairwise alignment (legacy align(), DW_banded.c:115-330): every pair is a
// two-sequence "pile" (target, query) whose window is forced to the full
// strings; only the alignment stage runs.  The gapped strings are expanded on
// the host from the device edit script (formatting of a device result).
// ---------------------------------------------------------------------------
extern "C" int fa_align_pairs(fa_ctx *ctx, int n, const char *const *q, const int *q_len,
                              const char *const *t, const int *t_len, int band_tolerance,
                              int get_aln_str, alignment **out) {
    if (n <= 0) return 0;
    // up to 190: the tuned kernel (k_align.hip); beyond, e.g. the 1500 of contig layout
    // (graph_to_contig.py:52-105): the general one (k_align_wide.hip)
    const bool wide = band_tolerance + 1 > 64 * FA_ALIGN_MAXCH - 1;
    if (band_tolerance < 0 || band_tolerance > FA_WIDE_BAND_MAX) {
        set_err("falcon_amd: band_tolerance %d unsupported by the GPU alignment kernels (max %d)",
                band_tolerance, FA_WIDE_BAND_MAX);
        return -1;
    }
    std::vector<const char *> seqs(2 * (size_t)n);
    std::vector<int> lens(2 * (size_t)n), pn((size_t)n, 2);
    for (int i = 0; i < n; i++) {
        seqs[2 * i] = t[i];
        lens[2 * i] = t_len[i];
        seqs[2 * i + 1] = q[i];
        lens[2 * i + 1] = q_len[i];
    }
    fa_batch *b = batch_build(ctx, n, pn.data(), seqs.data(), lens.data(), true, band_tolerance);
    if (!b) return -1;
    int rc = 0;
    fa_ctx *c = ctx;
    std::lock_guard<std::mutex> front(c->front_mu);
    hipStream_t s = c->stream;
    std::vector<FaRange> rg(b->n_seq);
    int max_q = 0, max_t = 0;
    for (int i = 0; i < n; i++) {
        FaRange z;
        memset(&z, 0, sizeof(z));
        rg[2 * i] = z;
        z.e1 = q_len[i];
        z.e2 = t_len[i];
        z.ok = 1;
        rg[2 * i + 1] = z;
        max_q = std::max(max_q, q_len[i]);
        max_t = std::max(max_t, t_len[i]);
    }
    auto fail = [&](int code) {
        fa_batch_free(b);
        return code;
    };
    b->front_launched = true;
    if (hipMemcpyAsync(b->d_range.p, rg.data(), rg.size() * sizeof(FaRange), hipMemcpyHostToDevice,
                       s) != hipSuccess) {
        set_err("falcon_amd: range upload failed");
        return fail(-1);
    }
    size_t lds = fa_align_lds_bytes(max_q, max_t);
    if (lds > 160 * 1024) {
        set_err("falcon_amd: sequences too long for the LDS-staged alignment kernel");
        return fail(-1);
    }
    trace_stage(s, "pair-range");
    FaBatchDev d = b->dev();
    b->stats.align_relaunched = 0;
    if (use_align2(b, band_tolerance)) {
        if (ensure_arena_a2(c, b)) return fail(-1);
        for (const auto &gr : b->a2_groups)
            fa_launch_align2(d, c->a2, 2.0, band_tolerance, b->d_order.p + gr.order_begin, gr.order_count, gr.word_base, s);
    } else {
        if (ensure_arena(c, b, lds, true)) return fail(-1);
        if (wide) fa_launch_align_wide(d, c->arena, 2.0, band_tolerance, s);
        else fa_launch_align_band(d, c->arena, max_q, max_t, 2.0, band_tolerance, s);
    }
    trace_stage(s, "pair-align");
    if (hipGetLastError() != hipSuccess) {
        set_err("falcon_amd: k_align launch failed");
        return fail(-1);
    }
    rc = fetch_aln(b, s);
    if (rc == 1) rc = redo_handed_back(b, 2.0, band_tolerance, s);
    if (rc) return fail(rc);
    std::vector<u32> script;
    if (get_aln_str > 0) {
        script.resize(b->script_words + 8);
        if (hipMemcpy(script.data(), b->d_script.p, b->script_words * sizeof(u32),
                      hipMemcpyDeviceToHost) != hipSuccess) {
            set_err("falcon_amd: script download failed");
            return fail(-1);
        }
    }
    for (int i = 0; i < n; i++) {
        const FaAln &a = b->h_aln[2 * i + 1];
        if (trace_on()) {
            fprintf(stderr, "[falcon_amd] pair %d: aligned %d dist %d q_e %d t_e %d size %d cells %lld err %d",
                    i, a.aligned, a.dist, a.q_e, a.t_e, a.size, a.cells, a.err);
            if (get_aln_str > 0)
                for (int k = 0; k < 4; k++) fprintf(stderr, " s[%d]=%u", k, script[b->script_off[2 * i + 1] + k]);
            fprintf(stderr, "\n");
        }
        alignment *r = (alignment *)calloc(1, sizeof(alignment));
        size_t cap = (size_t)q_len[i] + (size_t)t_len[i] + 1;
        r->q_aln_str = (char *)calloc(cap, 1);
        r->t_aln_str = (char *)calloc(cap, 1);
        if (a.aligned) {
            r->aln_str_size = a.size;
            r->dist = a.dist;
            r->aln_q_e = a.q_e;
            r->aln_t_e = a.t_e;
            if (get_aln_str > 0) {
                const u32 *sc = script.data() + b->script_off[2 * i + 1];
                int x = 0, y = 0;
                size_t pos = 0;
                for (int dd = 0; dd <= a.dist; dd++) {
                    u32 e = sc[dd];
                    if (dd > 0) {
                        if (e & 1u) {
                            r->q_aln_str[pos] = '-';
                            r->t_aln_str[pos] = t[i][y++];
                        } else {
                            r->q_aln_str[pos] = q[i][x++];
                            r->t_aln_str[pos] = '-';
                        }
                        pos++;
                    }
                    for (u32 m = e >> 1; m > 0; m--) {
                        r->q_aln_str[pos] = q[i][x++];
                        r->t_aln_str[pos] = t[i][y++];
                        pos++;
                    }
                }
                r->aln_str_size = (int)pos;
            }
        }
        out[i] = r;
    }
    fa_batch_free(b);
    return 0;
}
