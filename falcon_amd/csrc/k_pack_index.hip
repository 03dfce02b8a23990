// k_pack_index.hip -- (1) ASCII -> 2-bit packing, (2) per-pile seed 8-mer index.
//
// Replaces, for one whole batch of piles at once:
//   * the per-call ASCII->code loops of the reference
//     (src/c/kmer_lookup.c:159-171, :236-249), and
//   * add_sequence() (src/c/kmer_lookup.c:140-192): instead of the start/last
//     table plus a "next occurrence" chain we build a CSR table
//     kidx[kmer] .. kidx[kmer+1] of ascending seed positions, which enumerates
//     the same list in the same order (positions 0 .. seed_len-K-1, :174).
//
// Both kernels are HBM streaming work (byte in / quarter-byte out; 65537-entry
// table per pile); no MFMA.
#include "fa_device.h"

// --------------------------------------------------------------------------
// pack: one thread produces one u32 (16 bases) from one aligned 16-byte load.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ ascii,
                                              const u64 *__restrict__ ascii_off,
                                              const FaSeq *__restrict__ seq, int n_seq,
                                              u32 *__restrict__ words, u64 n_words,
                                              int *__restrict__ first_bad, int *__restrict__ bad_pile) {
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    // which sequence owns word w: last g with seq[g].woff <= w
    int lo = 0, hi = n_seq - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if ((u64)seq[mid].woff <= w) lo = mid; else hi = mid - 1;
    }
    const FaSeq s = seq[lo];
    u32 wi = (u32)(w - s.woff);
    int base0 = (int)wi * 16;
    u32 out = 0;
    bool bad = false;
    if (base0 < s.len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ascii + ascii_off[lo] + (u64)base0);
        const u32 q[4] = {v.x, v.y, v.z, v.w};
        int valid = min(16, s.len - base0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u32 c = (q[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            // 'A'->0 'C'->1 'G'->2 'T'->3
            u32 code = ((c >> 1) ^ (c >> 2)) & 3u;
            if (j < valid) {
                out |= code << (2 * j);
                // anything but upper-case ACGT is outside the parity domain: the reference
                // aligns raw characters and codes other bytes 0xff in the seed index and 0
                // in reads (kmer_lookup.c:159-171, :236-249) -- refuse it, never fold it
                const bool acgt = (c & 0xE0u) == 0x40u && ((0x0010008Au >> (c & 31u)) & 1u);
                bad |= !acgt;
            }
        }
    }
    words[w] = out;
    if (bad) {
        atomicMin(first_bad, lo);
        if (bad_pile) bad_pile[s.pile] = 1;  // (piles fail alone: the engine takes this one out)
    }
}

void fa_launch_pack(const FaBatchDev &b, int *first_bad, int *bad_pile, hipStream_t s) {
    if (b.n_words == 0) return;
    unsigned grid = (unsigned)((b.n_words + 255) / 256);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, s, b.ascii, b.ascii_off, b.seq, b.n_seq,
                       b.words, b.n_words, first_bad, bad_pile);
}

// --------------------------------------------------------------------------
// seed index: one 1024-thread workgroup per pile.
//   phase A  zero the table
//   phase B  histogram of the seed's 8-mers (global atomics, L2)
//   phase C  exclusive scan (tile of 1024 entries per step, LDS carry)
//   phase D  ordered fill by wave 0: 64 positions per step in ascending order;
//            lanes holding the same 8-mer are found with 16 ballots, ranked by
//            lane id, and the group leader reserves the slots with one atomic,
//            so every bucket ends up in ascending position order
//            (== the reference chain order) without any sort.
// After phase D, T[k+1] (the bucket cursor) equals the bucket end, and T[0]=0,
// so bucket(k) = [T[k], T[k+1]).
// --------------------------------------------------------------------------
#define SI_NT 512  // threads per pile: few, fat workgroups keep the tables in flight (262 KB
                    // each) within the 256 MB Infinity Cache; 256-thread ones put 537 MB in flight
__global__ __launch_bounds__(SI_NT) void k_seed_index(const u32 *__restrict__ words,
                                                    const FaSeq *__restrict__ seq,
                                                    const FaPile *__restrict__ pile,
                                                    u32 *__restrict__ kidx,
                                                    u32 *__restrict__ kpos) {
    const FaPile pm = pile[blockIdx.x];
    const FaSeq sd = seq[pm.first];
    const u32 *w = words + sd.woff;
    u32 *T = kidx + pm.kidx_off;   // 65537 entries used
    u32 *P = kpos + pm.kpos_off;
    const int tid = threadIdx.x;
    const int n_pos = max(0, sd.len - FA_K);

    for (int i = tid; i < FA_NKMER + 1; i += SI_NT) T[i] = 0;
    __syncthreads();

    for (int i = tid; i < n_pos; i += SI_NT) atomicAdd(&T[fa_kmer8(w, i) + 1], 1u);
    __syncthreads();

    // exclusive scan of T[1..65536] in place: tiles of 4 entries per thread; DPP scan
    // inside each wavefront, the wavefront totals through LDS (one barrier per tile)
    constexpr int NW = SI_NT / 64;
    __shared__ u32 s_wtot[2][NW];
    u32 carry = 0;  // sum of all earlier tiles (kept by every thread)
    for (int tile = 0, ph = 0; tile < FA_NKMER; tile += 4 * SI_NT, ph ^= 1) {
        u32 v[4];
        // T+1 is only 4-byte aligned, so read scalars
        v[0] = T[1 + tile + tid * 4 + 0];
        v[1] = T[1 + tile + tid * 4 + 1];
        v[2] = T[1 + tile + tid * 4 + 2];
        v[3] = T[1 + tile + tid * 4 + 3];
        const u32 sum = v[0] + v[1] + v[2] + v[3];
        u32 inc = sum;  // inclusive scan over the 64 lanes (row_shr 1,2,4,8, row_bcast 15/31)
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);
        inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);
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

void fa_launch_index(const FaBatchDev &b, hipStream_t s) {
    if (b.n_pile == 0) return;
    hipLaunchKernelGGL(k_seed_index, dim3(b.n_pile), dim3(SI_NT), 0, s, b.words, b.seq, b.pile,
                       b.kidx, b.kpos);
}

// (fa_warm: the code object of this file is loaded when one of its kernels is first looked at)
void fa_touch_index() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_seed_index));
}
