// tests/emu/simt/simt.cpp -- TEST INFRASTRUCTURE: fibers and the rendezvous scheduler of simt.h.
#include "simt.h"

#include <sys/mman.h>

namespace simt {
Wave g_wave;
Wave *g_cw = &g_wave;
int g_nwaves = 1;
void *g_dyn_lds = nullptr;

// x86-64 SysV stack switch: callee-saved registers on the old stack, stack pointers swapped.
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size simt_switch,.-simt_switch
)");

static constexpr size_t STACK = 256 * 1024;

static void fiber_main() {
    Wave &w = *g_cw;
    w.body();
    Fiber &me = w.f[w.cur];
    me.done = true;
    simt_switch(&me.sp, w.sched_sp);
    abort();  // a finished fiber is never resumed
}

static void prepare(Fiber &f) {
    if (!f.stack) {
        f.stack = (char *)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (f.stack == (char *)MAP_FAILED) { perror("simt: mmap"); abort(); }
    }
    // frame simt_switch pops: r15 r14 r13 r12 rbx rbp, then `ret` into fiber_main with the
    // stack as after a call (rsp % 16 == 8)
    uint64_t *top = (uint64_t *)(f.stack + STACK);
    top -= 2;               // keep 16 bytes above
    *--top = 0;             // fake return address of fiber_main (never used)
    *--top = (uint64_t)(void *)&fiber_main;
    for (int i = 0; i < 6; i++) *--top = 0;
    f.sp = top;
    f.done = false;
    f.waiting = false;
    f.site = 0;
}

// Runs the lanes of wavefront w until every live one waits: returns the site they wait at
// (-1: all lanes have returned), the live lanes in `act`.
static int run_to_rendezvous(Wave &w, uint64_t &act) {
    g_cw = &w;
    for (int l = 0; l < W; l++) {
        Fiber &f = w.f[l];
        if (f.done || f.waiting) continue;
        w.cur = l;
        simt_switch(&w.sched_sp, f.sp);
    }
    act = 0;
    int site = -1;
    for (int l = 0; l < W; l++) {
        Fiber &f = w.f[l];
        if (f.done) continue;
        if (site < 0) site = f.site;
        if (f.site != site) {
            fprintf(stderr, "simt: %s block %u wavefront %d: lanes wait at different cross-lane sites (line %d and lane %d at "
                            "line %d): a cross-lane operation sits in divergent control flow\n",
                    w.kernel, w.block, w.index, site & ~BLOCK_SITE, l, f.site & ~BLOCK_SITE);
            abort();
        }
        act |= 1ull << l;
    }
    return act ? site : -1;
}

static void release(Wave &w, uint64_t act) {
    w.act[w.gen & 1u] = act;
    w.gen++;
    g_wave.n_sync++;
    for (int l = 0; l < W; l++) w.f[l].waiting = false;
}

static Wave g_more[MAX_WAVES - 1];

void run_block_waves(const char *name, unsigned block, unsigned grid, int n_waves, const std::function<void()> &body) {
    if (n_waves < 1 || n_waves > MAX_WAVES) { fprintf(stderr, "simt: %d wavefronts per block\n", n_waves); abort(); }
    Wave *ws[MAX_WAVES];
    uint64_t act[MAX_WAVES];
    bool finished[MAX_WAVES];
    g_nwaves = n_waves;
    for (int i = 0; i < n_waves; i++) {
        Wave &w = *(ws[i] = i == 0 ? &g_wave : &g_more[i - 1]);
        w.kernel = name; w.block = block; w.grid = grid; w.body = body; w.index = i;
        w.gen = 0; w.at_barrier = false;
        finished[i] = false;
        for (int l = 0; l < W; l++) prepare(w.f[l]);
    }
    for (;;) {
        int live = 0;
        for (int i = 0; i < n_waves; i++) {
            Wave &w = *ws[i];
            while (!finished[i] && !w.at_barrier) {  // this wavefront, until it returns or meets a block barrier
                const int site = run_to_rendezvous(w, act[i]);
                if (site < 0) finished[i] = true;
                else if (site & BLOCK_SITE) w.at_barrier = true;
                else release(w, act[i]);
            }
            if (!finished[i]) live++;
        }
        if (live == 0) break;
        // (every live wavefront waits at a block barrier: the sites may differ -- wavefronts of
        // different roles call __syncthreads() from different lines)
        for (int i = 0; i < n_waves; i++)
            if (!finished[i]) { ws[i]->at_barrier = false; release(*ws[i], act[i]); }
    }
    g_cw = &g_wave;
}

void run_block(Wave &w) {
    (void)w;
    run_block_waves(g_wave.kernel, g_wave.block, g_wave.grid, 1, g_wave.body);
}
}  // namespace simt
