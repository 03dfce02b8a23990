// tests/emu/simt/simt.cpp -- TEST INFRASTRUCTURE: fibers and the rendezvous scheduler of simt.h.
#include "simt.h"

#include <sys/mman.h>

namespace simt {
Wave g_wave;
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
    Wave &w = g_wave;
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

void run_block(Wave &w) {
    for (int l = 0; l < W; l++) prepare(w.f[l]);
    w.gen = 0;
    for (;;) {
        bool any = false;
        for (int l = 0; l < W; l++) {
            Fiber &f = w.f[l];
            if (f.done || f.waiting) continue;
            w.cur = l;
            simt_switch(&w.sched_sp, f.sp);
        }
        uint64_t act = 0;
        int site = -1;
        for (int l = 0; l < W; l++) {
            Fiber &f = w.f[l];
            if (f.done) continue;
            any = true;
            if (site < 0) site = f.site;
            if (f.site != site) {
                fprintf(stderr, "simt: %s block %u: lanes wait at different cross-lane sites (line %d and lane %d at line %d): "
                                "a cross-lane operation sits in divergent control flow\n",
                        w.kernel, w.block, site, l, f.site);
                abort();
            }
            act |= 1ull << l;
        }
        if (!any) break;
        w.act[w.gen & 1u] = act;
        w.gen++;
        w.n_sync++;
        for (int l = 0; l < W; l++) w.f[l].waiting = false;
    }
}
}  // namespace simt
