// k_seed_index.hip -- the seed's 8-mer index (kmer_lookup.c:71-192: allocate_kmer_lookup,
// init_kmer_lookup, add_sequence), one workgroup per pile, THE TABLE IN THE LDS.
//
// The index of a pile is a CSR table: T[k] .. T[k+1] bounds the positions of 8-mer k in P, in
// ascending position order (the order the reference's per-k-mer chains are walked in,
// kmer_lookup.c:141-170).  Rounds 1-3 built it in global memory -- histogram with global atomics,
// scan, ordered fill with returning atomics: 1024 tables of 262 KB in flight are 268 MB of randomly
// touched lines, nothing of which an XCD's 4 MB L2 holds, and every one of the 3 x 61 M scattered
// accesses of a batch moved a line in and out (6.3 ms per 3072 piles for 0.24 G instructions;
// that kernel is k_seed_index_long below, kept for seeds of 65 544 bases and more).
//
// Here the 65 536 counters are 16 bits wide and live in 128 KB of LDS (a seed of < 65 544 bases has
// at most 65 535 positions, so no counter and no cursor overflows):
//   A  zero the table; the seed's packed bases (<= 16.4 KB) into the LDS as well, so that neither the
//      histogram nor the fill waits for global memory
//   B  histogram: ds_add of 1 << 16*(k & 1) on word k >> 1
//   C  exclusive scan: a wavefront owns 2048 consecutive words, a lane 4 of them per step (one
//      ds_read_b128), eight steps in registers; DPP scans inside the steps, the wavefronts' totals
//      through the LDS; the bounds go to T (two 16-byte stores per lane and step) and back into
//      the table as the buckets' cursors
//   D  ordered fill by wavefront 0, 64 positions per step in ascending order: read the cursor, add
//      one, read it again -- LDS operations of a wavefront execute in order, so a lane that sees
//      exactly its own increment was alone with its 8-mer in this step and the first read IS its
//      slot.  Only a step in which some 8-mer occurs twice (tandem repeats, low complexity) ranks
//      its lanes with the 16 ballots of round 1.
// P is written straight to global memory: 4 bytes per position into an 80 KB region per pile, which
// the L2 does hold.
#include <hip/hip_runtime.h>

#include "fa_device.h"
#include "k_msa.h"

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define SI_NT 1024
#define SI_NW (SI_NT / 64)
#define SI_WORDS (FA_NKMER / 2)    // 32768 words of two 16-bit counters
#define SI_WPW (SI_WORDS / SI_NW)  // words per wavefront in the scan
#define SI_STEPS (SI_WPW / 256)    // steps of 64 lanes x 4 words
#define SI_MAX_POS 65535
#define SI_SEED_WORDS ((SI_MAX_POS + FA_K + 15) / 16 + 2)  // packed words of the longest seed + the window's second word

// the table word of 8-mer k, and the increment / shift of its half
#define SI_WORD(k) ((k) >> 1)
#define SI_SHIFT(k) (((k) & 1u) * 16u)

// phases A + B: zero, histogram (tid: thread of the workgroup)
__device__ __forceinline__ void si_zero(u32 *cur, u32 *sw, const u32 *w, int len, int tid) {
    const int n_words = (len + 15) / 16 + 1;  // (two zero words follow every sequence: fa_internal.h)
    for (int i = tid; i < n_words; i += SI_NT) sw[i] = w[i];
    for (int i = tid * 4; i < SI_WORDS; i += SI_NT * 4) *(u32x4 *)&cur[i] = (u32x4)(0u);
}
__device__ __forceinline__ void si_histogram(u32 *cur, const u32 *w, int n_pos, int tid) {
    for (int i = tid; i < n_pos; i += SI_NT) {
        const u32 k = fa_kmer8(w, i);
        atomicAdd(&cur[SI_WORD(k)], 1u << SI_SHIFT(k));
    }
}

// phase C, first half: the lane's 32 words into registers, the inclusive scan over the lanes of
// every step, the wavefront's total
struct SiScan {
    u32x4 v[SI_STEPS];
    int incl[SI_STEPS], sum[SI_STEPS];
};
__device__ __forceinline__ int si_scan_load(const u32 *cur, int wv, int lane, SiScan &S) {
    int total = 0;
#pragma unroll
    for (int s = 0; s < SI_STEPS; s++) {
        S.v[s] = *(const u32x4 *)&cur[wv * SI_WPW + s * 256 + lane * 4];
        u32 t = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) t += (S.v[s][j] & 0xffffu) + (S.v[s][j] >> 16);
        S.sum[s] = (int)t;
    }
#pragma unroll
    for (int s = 0; s < SI_STEPS; s++) {
        S.incl[s] = wave_incl_sum(S.sum[s], lane);
        total += __builtin_amdgcn_readlane(S.incl[s], 63);
    }
    return total;
}
// second half: bounds to T, cursors back into the table (carry: entries below this wavefront)
__device__ __forceinline__ void si_scan_store(u32 *cur, u32 *T, int wv, int lane, const SiScan &S, u32 carry) {
#pragma unroll
    for (int s = 0; s < SI_STEPS; s++) {
        const int word = wv * SI_WPW + s * 256 + lane * 4;
        u32 a = carry + (u32)(S.incl[s] - S.sum[s]);
        u32x4 t0, t1, c;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 lo = S.v[s][j] & 0xffffu, hi = S.v[s][j] >> 16;
            const u32 a1 = a + lo;
            c[j] = a | (a1 << 16);
            if (j < 2) {
                t0[2 * j] = a;
                t0[2 * j + 1] = a1;
            } else {
                t1[2 * (j - 2)] = a;
                t1[2 * (j - 2) + 1] = a1;
            }
            a = a1 + hi;
        }
        *(u32x4 *)&T[2 * word] = t0;
        *(u32x4 *)&T[2 * word + 4] = t1;
        *(u32x4 *)&cur[word] = c;
        carry += (u32)__builtin_amdgcn_readlane(S.incl[s], 63);
    }
}

// phase D (one wavefront): positions in ascending order into their buckets
__device__ __forceinline__ void si_fill(u32 *cur, const u32 *w, u32 *P, int n_pos, int lane) {
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr int U = 4;
    for (int i0 = 0; i0 < n_pos; i0 += 64 * U) {
        u32 km[U], pre[U], post[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + 64 * u + lane;
            valid[u] = i < n_pos;
            km[u] = fa_kmer8(w, min(i, n_pos - 1));  // (no branch around the read; a lane beyond the end adds 0)
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u32 *word = &cur[SI_WORD(km[u])];
            // (relaxed atomic loads: the compiler may neither move them across the add nor
            // derive the second from the first)
            pre[u] = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            fa_lds_order();
            atomicAdd(word, valid[u] ? (1u << SI_SHIFT(km[u])) : 0u);
            fa_lds_order();
            post[u] = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 a = (pre[u] >> SI_SHIFT(km[u])) & 0xffffu, b = (post[u] >> SI_SHIFT(km[u])) & 0xffffu;
            u32 slot = a;
            const bool crowded = valid[u] && (b - a != 1u);
            if (fa_ballot(crowded) != 0) {  // (wave-uniform) some 8-mer twice among these 64 positions
                u64 peers = fa_ballot(valid[u]);
#pragma unroll
                for (int bit = 0; bit < 16; bit++) {
                    const bool one = (km[u] >> bit) & 1u;
                    const u64 bal = fa_ballot(one);
                    peers &= one ? bal : ~bal;
                }
                slot = a + (u32)__popcll(peers & lt_mask);
            }
            if (valid[u]) P[slot] = (u32)(i0 + 64 * u + lane);
        }
    }
}

// One pile, thread `tid` of its workgroup.
__device__ __forceinline__ void si_pile(u32 *cur, u32 *wtot, u32 *sw, const u32 *seed_words, int len, u32 *T, u32 *P,
                                        int tid) {
    const int n_pos = max(0, len - FA_K), lane = tid & 63, wv = tid >> 6;
    si_zero(cur, sw, seed_words, len, tid);
    __syncthreads();
    si_histogram(cur, sw, n_pos, tid);
    __syncthreads();
    SiScan S;
    const int total = si_scan_load(cur, wv, lane, S);
    if (lane == 0) wtot[wv] = (u32)total;
    __syncthreads();
    u32 carry = 0;
#pragma unroll
    for (int q = 0; q < SI_NW; q++) carry += q < wv ? wtot[q] : 0u;
    si_scan_store(cur, T, wv, lane, S, carry);
    if (tid == 0) T[FA_NKMER] = (u32)n_pos;
    __syncthreads();
    if (wv == 0) si_fill(cur, sw, P, n_pos, lane);
}

__global__ __launch_bounds__(SI_NT) void k_seed_index(const u32 *__restrict__ words, const FaSeq *__restrict__ seq,
                                                      const FaPile *__restrict__ pile, u32 *__restrict__ kidx,
                                                      u32 *__restrict__ kpos) {
    __shared__ __attribute__((aligned(16))) u32 cur[SI_WORDS];
    __shared__ u32 wtot[SI_NW];
    __shared__ u32 sw[SI_SEED_WORDS];
    const FaPile pm = pile[blockIdx.x];
    const FaSeq sd = seq[pm.first];
    if (sd.len - FA_K > SI_MAX_POS) return;  // k_seed_index_long's
    si_pile(cur, wtot, sw, words + sd.woff, sd.len, kidx + pm.kidx_off, kpos + pm.kpos_off, threadIdx.x);
}

// --------------------------------------------------------------------------
// Seeds of 65 544 .. 99 999 bases: the table in global memory, 32-bit counters.
//   phase A  zero the table
//   phase B  histogram of the seed's 8-mers (global atomics, L2)
//   phase C  exclusive scan (tile of 4 entries per thread per step, LDS carry)
//   phase D  ordered fill by wave 0: 64 positions per step in ascending order;
//            lanes holding the same 8-mer are found with 16 ballots, ranked by
//            lane id, and the group leader reserves the slots with one atomic,
//            so every bucket ends up in ascending position order without any sort.
// After phase D, T[k+1] (the bucket cursor) equals the bucket end, and T[0]=0,
// so bucket(k) = [T[k], T[k+1]).
// --------------------------------------------------------------------------
#define SL_NT 512
__global__ __launch_bounds__(SL_NT) void k_seed_index_long(const u32 *__restrict__ words,
                                                           const FaSeq *__restrict__ seq,
                                                           const FaPile *__restrict__ pile,
                                                           u32 *__restrict__ kidx, u32 *__restrict__ kpos,
                                                           int every_pile) {
    const FaPile pm = pile[blockIdx.x];
    const FaSeq sd = seq[pm.first];
    const u32 *w = words + sd.woff;
    u32 *T = kidx + pm.kidx_off;
    u32 *P = kpos + pm.kpos_off;
    const int tid = threadIdx.x;
    const int n_pos = max(0, sd.len - FA_K);
    if (n_pos <= SI_MAX_POS && !every_pile) return;  // k_seed_index's

    for (int i = tid; i < FA_NKMER + 1; i += SL_NT) T[i] = 0;
    __syncthreads();

    for (int i = tid; i < n_pos; i += SL_NT) atomicAdd(&T[fa_kmer8(w, i) + 1], 1u);
    __syncthreads();

    // exclusive scan of T[1..65536] in place: tiles of 4 entries per thread; DPP scan
    // inside each wavefront, the wavefront totals through LDS (one barrier per tile)
    constexpr int NW = SL_NT / 64;
    __shared__ u32 s_wtot[2][NW];
    u32 carry = 0;  // sum of all earlier tiles (kept by every thread)
    for (int tile = 0, ph = 0; tile < FA_NKMER; tile += 4 * SL_NT, ph ^= 1) {
        u32 v[4];
        // T+1 is only 4-byte aligned, so read scalars
        v[0] = T[1 + tile + tid * 4 + 0];
        v[1] = T[1 + tile + tid * 4 + 1];
        v[2] = T[1 + tile + tid * 4 + 2];
        v[3] = T[1 + tile + tid * 4 + 3];
        const u32 sum = v[0] + v[1] + v[2] + v[3];
        const u32 inc = (u32)wave_incl_sum((int)sum, tid & 63);
        const int wv = tid >> 6;
        if ((tid & 63) == 63) s_wtot[ph][wv] = inc;
        __syncthreads();  // (the other half of s_wtot is rewritten only after the next barrier)
        u32 below = 0, all = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const u32 wt = s_wtot[ph][q];
            below += q < wv ? wt : 0u;
            all += wt;
        }
        const u32 excl = carry + below + inc - sum;
        T[1 + tile + tid * 4 + 0] = excl;
        T[1 + tile + tid * 4 + 1] = excl + v[0];
        T[1 + tile + tid * 4 + 2] = excl + v[0] + v[1];
        T[1 + tile + tid * 4 + 3] = excl + v[0] + v[1] + v[2];
        carry += all;
    }
    __threadfence_block();
    __syncthreads();

    if (tid >= 64) return;
    const int lane = tid;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Eight steps of 64 positions at a time: their slot reservations (returning atomics,
    // a round trip to L2 each) are issued back to back and consumed afterwards.  Atomics
    // of one wavefront on one address execute in program order, so buckets still fill in
    // ascending position order.
    constexpr int U = 8;
    for (int i0 = 0; i0 < n_pos; i0 += 64 * U) {
        u32 km[U], base[U];
        int rank[U], leader[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + 64 * u + lane;
            valid[u] = i < n_pos;
            km[u] = valid[u] ? fa_kmer8(w, i) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 peers = __ballot(valid[u]);
#pragma unroll
            for (int bit = 0; bit < 16; bit++) {
                const bool one = (km[u] >> bit) & 1u;
                const u64 bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            if (!valid[u]) peers = 0;
            rank[u] = __popcll(peers & lt_mask);
            const int cnt = __popcll(peers);
            leader[u] = valid[u] ? (__ffsll((long long)peers) - 1) : lane;
            base[u] = 0;
            if (valid[u] && lane == leader[u]) base[u] = atomicAdd(&T[km[u] + 1], (u32)cnt);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 b = (u32)__shfl((int)base[u], leader[u]);
            if (valid[u]) P[b + (u32)rank[u]] = (u32)(i0 + 64 * u + lane);
        }
    }
}

#ifndef FA_EMU
// max_seed_len: the batch's longest seed (the kernel behind is only launched when a seed needs it);
// FALCON_AMD_INDEX_LONG=1: every pile through the kernel behind (tests, A/B runs)
void fa_launch_index(const FaBatchDev &b, int max_seed_len, hipStream_t s) {
    if (b.n_pile == 0) return;
    const bool every_pile = getenv("FALCON_AMD_INDEX_LONG") != nullptr;  // (read per launch, on the submitting thread)
    if (!every_pile)
        hipLaunchKernelGGL(k_seed_index, dim3(b.n_pile), dim3(SI_NT), 0, s, b.words, b.seq, b.pile, b.kidx, b.kpos);
    if (every_pile || max_seed_len - FA_K > SI_MAX_POS)
        hipLaunchKernelGGL(k_seed_index_long, dim3(b.n_pile), dim3(SL_NT), 0, s, b.words, b.seq, b.pile, b.kidx,
                           b.kpos, every_pile ? 1 : 0);
}

// (fa_warm: the code object of this file is loaded when one of its kernels is first looked at)
void fa_touch_index() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_seed_index));
}
#endif  // FA_EMU
