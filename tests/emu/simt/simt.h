// tests/emu/simt/simt.h -- TEST INFRASTRUCTURE: a wave64 SIMT emulator for the host.
//
// Runs HIP kernel SOURCE written in the ordinary per-thread style (k_msa.hip, k_score2.hip)
// on the CPU: the 64 lanes of a wavefront are 64 fibers on one OS thread, every cross-lane
// operation (ballot, shuffle, DPP, readlane, ds_permute / ds_bpermute, the wave barrier) is a
// rendezvous -- each lane deposits its operand and yields, and when every live lane has
// arrived at the SAME call site the lanes resume and compute their results from the deposited
// operands.  A lane that returns from the kernel early drops out of later rendezvous (its
// ballot bit reads 0), like a lane whose exec bit is cleared for good.
//
// What the model demands of the kernel source (checked where it can be):
//   * cross-lane operations sit in wave-uniform control flow: lanes waiting at different call
//     sites abort the run with both sites named;
//   * lanes that talk through memory (LDS or global) put a rendezvous between the write and
//     the read -- fa_wave_sync() where the hardware needs nothing because the lanes run in
//     lockstep; __syncthreads() and __threadfence_block() are rendezvous as well.
// Blocks run one after the other (a `__shared__` array is a function-local static).  A block of
// several wavefronts (launch_waves) runs them one after the other as well: each until it has
// returned or every live lane of it waits at a __syncthreads(), which lets go when all the block's
// live wavefronts are there (a wavefront that has returned is not waited for, as on the hardware).
// Never compiled into the product; nothing under falcon_amd/ includes it.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

namespace simt {
constexpr int W = 64;
struct Dim3 { unsigned x, y, z; };

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = true, waiting = false;
    int site = 0;
};

struct Wave {
    Fiber f[W];
    void *sched_sp = nullptr;
    int cur = 0;
    uint64_t slot[2][W];
    uint64_t act[2];
    unsigned gen = 0;
    unsigned block = 0, grid = 1;
    std::function<void()> body;
    const char *kernel = "";
    unsigned long long n_sync = 0;
    int index = 0;            // wavefront of its block
    bool at_barrier = false;  // every live lane waits at a __syncthreads()
};
constexpr int MAX_WAVES = 16;
constexpr int BLOCK_SITE = 1 << 30;  // site flag of a rendezvous of the whole block
extern Wave g_wave;            // wavefront 0 (and the counters the drivers read)
extern Wave *g_cw;             // the wavefront that is running
extern int g_nwaves;           // wavefronts per block of the running launch

extern "C" void simt_switch(void **save_sp, void *load_sp);
void run_block(Wave &w);
void run_block_waves(const char *name, unsigned block, unsigned grid, int n_waves, const std::function<void()> &body);

// dynamic `extern __shared__` memory of the running block (the launcher sets it)
extern void *g_dyn_lds;
inline int lane() { return g_cw->cur; }
inline Dim3 tidx() { return Dim3{(unsigned)(g_cw->index * W + g_cw->cur), 0, 0}; }
inline Dim3 bidx() { return Dim3{g_cw->block, 0, 0}; }
inline Dim3 gdim() { return Dim3{g_cw->grid, 1, 1}; }
inline Dim3 bdim() { return Dim3{(unsigned)(W * g_nwaves), 1, 1}; }

struct X {
    const uint64_t *v;
    uint64_t act;
};
// deposit `mine`, wait for the other live lanes, return everybody's operands
inline X xchg(uint64_t mine, int site) {
    Wave &w = *g_cw;
    const int me = w.cur;
    const unsigned g = w.gen & 1u;
    w.slot[g][me] = mine;
    w.f[me].site = site;
    w.f[me].waiting = true;
    simt_switch(&w.f[me].sp, w.sched_sp);
    return X{w.slot[g], w.act[g]};
}

// ---- cross-lane primitives -------------------------------------------------------------
inline uint64_t ballot(bool p, int site) {
    const X x = xchg(p ? 1u : 0u, site);
    uint64_t m = 0;
    for (int l = 0; l < W; l++)
        if (((x.act >> l) & 1ull) && x.v[l]) m |= 1ull << l;
    return m;
}
// (32- and 64-bit operands alike: a slot holds 64 bits; a lane that left reads as 0)
template <class T> inline T shfl(T v, int src, int site) {
    static_assert(sizeof(T) <= 8, "shfl operand");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    const X x = xchg(bits, site);
    const uint64_t r = ((x.act >> (src & 63)) & 1ull) ? x.v[src & 63] : 0;
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T> inline T shfl_up(T v, int delta, int site) {
    const int me = lane();
    const T r = shfl(v, me >= delta ? me - delta : me, site);
    return me >= delta ? r : v;
}
template <class T> inline T shfl_xor(T v, int mask, int site) { return shfl(v, lane() ^ mask, site); }
inline int readlane(int v, int l, int site) {
    const X x = xchg((uint32_t)v, site);
    return (int)(uint32_t)x.v[l & 63];
}
inline int readfirstlane(int v, int site) {
    const X x = xchg((uint32_t)v, site);
    return (int)(uint32_t)x.v[__builtin_ctzll(x.act)];
}
inline void sync(int site) { (void)xchg(0, site); }
inline void sync_block(int site) { (void)xchg(0, site | BLOCK_SITE); }

// DPP: which lane does `me` read under control `ctrl` (-1: no source lane)
inline int dpp_src(int me, int ctrl) {
    const int row = me >> 4, col = me & 15;
    if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl - 0x100; return col + n < 16 ? me + n : -1; }  // row_shl:n
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl - 0x110; return col >= n ? me - n : -1; }     // row_shr:n
    if (ctrl == 0x130) return me < 63 ? me + 1 : -1;   // wave_shl:1
    if (ctrl == 0x138) return me > 0 ? me - 1 : -1;    // wave_shr:1
    if (ctrl == 0x142) return row >= 1 ? (row - 1) * 16 + 15 : -1;  // row_bcast:15
    if (ctrl == 0x143) return row >= 2 ? 31 : -1;                   // row_bcast:31
    fprintf(stderr, "simt: DPP control 0x%x is not modelled\n", ctrl);
    abort();
}
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site) {
    const X x = xchg((uint32_t)src, site);
    const int me = lane();
    if (!((row_mask >> (me >> 4)) & 1) || !((bank_mask >> ((me & 15) >> 2)) & 1)) return old;
    const int s = dpp_src(me, ctrl);
    if (s < 0 || !((x.act >> s) & 1ull)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)x.v[s];
}
inline int ds_bpermute(int addr, int v, int site) {
    const X x = xchg((uint32_t)v, site);
    return (int)(uint32_t)x.v[(addr >> 2) & 63];
}
inline int ds_permute(int addr, int v, int site) {
    const X x = xchg(((uint64_t)(uint32_t)addr << 32) | (uint32_t)v, site);
    const int me = lane();
    int r = 0;
    for (int l = 0; l < W; l++)
        if (((x.act >> l) & 1ull) && (int)((x.v[l] >> 34) & 63) == me) r = (int)(uint32_t)x.v[l];
    return r;
}
inline uint32_t mbcnt_lo(uint32_t m, uint32_t add) {
    const int me = lane();
    const uint32_t below = me >= 32 ? 0xffffffffu : ((1u << me) - 1u);
    return add + (uint32_t)__builtin_popcount(m & below);
}
inline uint32_t mbcnt_hi(uint32_t m, uint32_t add) {
    const int me = lane();
    const uint32_t below = me <= 32 ? 0u : ((1u << (me - 32)) - 1u);
    return add + (uint32_t)__builtin_popcount(m & below);
}

// ---- launching ---------------------------------------------------------------------------
template <class F>
inline void launch(const char *name, unsigned grid, F kernel_call) {
    const std::function<void()> body = kernel_call;
    for (unsigned b = 0; b < grid; b++) run_block_waves(name, b, grid, 1, body);
}
// blocks of n_waves wavefronts (threadIdx.x = 64 * wavefront + lane)
template <class F>
inline void launch_waves(const char *name, unsigned grid, int n_waves, F kernel_call) {
    const std::function<void()> body = kernel_call;
    for (unsigned b = 0; b < grid; b++) run_block_waves(name, b, grid, n_waves, body);
}
}  // namespace simt

// ---- the HIP names the kernels use ----------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static
#define threadIdx (simt::tidx())
#define blockIdx (simt::bidx())
#define gridDim (simt::gdim())
#define blockDim (simt::bdim())

#define __syncthreads() simt::sync_block(__LINE__)
#define __threadfence_block() simt::sync(__LINE__)
#define __ballot(p) simt::ballot((p), __LINE__)
#define __shfl(v, s) simt::shfl((v), (int)(s), __LINE__)
#define __shfl_up(v, d) simt::shfl_up((v), (int)(d), __LINE__)
#define __shfl_xor(v, m) simt::shfl_xor((v), (int)(m), __LINE__)
#define __clzll(x) __builtin_clzll((unsigned long long)(x))
#define __ffsll(x) __builtin_ffsll((long long)(x))
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#define __builtin_amdgcn_ballot_w64(p) simt::ballot((p), __LINE__)
#define __builtin_amdgcn_inverse_ballot_w64(m) ((bool)(((unsigned long long)(m) >> simt::lane()) & 1ull))
#define __builtin_amdgcn_readlane(v, l) simt::readlane((int)(v), (int)(l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) simt::readfirstlane((int)(v), __LINE__)
#define __builtin_amdgcn_update_dpp(o, s, c, rm, bm, bc) simt::update_dpp((int)(o), (int)(s), (c), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_mov_dpp(s, c, rm, bm, bc) simt::update_dpp(0, (int)(s), (c), (rm), (bm), (bc), __LINE__)
#define __builtin_amdgcn_ds_bpermute(a, v) simt::ds_bpermute((int)(a), (int)(v), __LINE__)
#define __builtin_amdgcn_ds_permute(a, v) simt::ds_permute((int)(a), (int)(v), __LINE__)
#define __builtin_amdgcn_mbcnt_lo(m, a) simt::mbcnt_lo((m), (a))
#define __builtin_amdgcn_mbcnt_hi(m, a) simt::mbcnt_hi((m), (a))
#define __builtin_amdgcn_wave_barrier() simt::sync(__LINE__)

#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __hip_atomic_load(p, order, scope) (*(p))
template <class T> inline T atomicAdd(T *p, T v) { const T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicMax(T *p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T *p, T v) { const T o = *p; if (v < o) *p = v; return o; }
using std::max;
using std::min;
inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
inline double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
struct uint2 { unsigned x, y; };
