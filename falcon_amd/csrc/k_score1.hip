// k_score1.hip -- the score recurrence, general kernel (see k_score2.hip for the one that runs by default).
#include "k_msa.h"

// ---------------------------------------------------------------------------
// k_score: one wavefront per pile, the score recurrence of falcon.c:405-475.
//
// The scores of the position being scored and of the one before it live in two
// VGPRs, lane = delta * 5 + base for the first SC_REG insertion levels (deeper
// levels are rare and take a generic path through LDS).  A level's links are
// decoded with lanes = links, then visited in stored order (= the reference's
// insertion order, Q5) with readlane: the previous node's score is a readlane of
// the score register, the node's accumulator is the lane it will be read from
// later, so the dependent chain of a level contains no memory access at all.
// Link words of a block of positions are staged in LDS with coalesced bursts and
// requested one level ahead; position records and link counts are handed out of
// registers; node records leave with one coalesced store per position.
// ---------------------------------------------------------------------------
#define SC_REG 12         // insertion levels whose scores live in registers
#define SC_ZERO 63        // lane of the score registers that always holds 0 (start links)

struct ScoreAcc { int h, p, k, n; };

// visit n_here links (lanes 0 .. n_here-1 of nbv / cv / lidxv / pidv); accumulators
// of node b are the lanes lane_base + b of acc.  FROM_CUR: predecessors are read
// from acc.h itself (delta >= 1), else from src_h (previous position).
template <bool FROM_CUR>
__device__ __forceinline__ void score_links(ScoreAcc &acc, int src_h, int nbv, int cv, int lidxv,
                                            int pidv, int n_here, int lane_rel) {
    for (int l = 0; l < n_here; l++) {
        const int nbl = __builtin_amdgcn_readlane(nbv, l);
        const int cl = __builtin_amdgcn_readlane(cv, l);
        const int li = __builtin_amdgcn_readlane(lidxv, l);
        const int pl = __builtin_amdgcn_readlane(pidv, l);
        const int ph = __builtin_amdgcn_readlane(FROM_CUR ? acc.h : src_h, li);
        const int h = ph + cl;  // falcon.c:440-445, half units
        const bool mine = lane_rel == nbl;
        const bool better = mine && h > acc.h;  // strict: first maximum (:447)
        acc.h = better ? h : acc.h;
        acc.p = better ? pl : acc.p;
        acc.k = better ? acc.n : acc.k;
        acc.n += mine ? 1 : 0;
    }
}

// ---- the unusual levels, out of line -----------------------------------------------
// More than 16 links, a level beyond the register-resident ones, or a predecessor there.
// Kept out of the hot loop as a call whose per-lane state travels through LDS (the caller
// parks its score lanes there and takes them back): inlined, or with the state passed and
// returned in registers, its control flow joins cost the fast path dozens of register
// copies per position.
struct ScoreSlowIo {
    int cur_h, cur_p, cur_k;   // score lanes of the position being scored
    int dg_h, dg_slot, dg_ck;  // best deep node so far (lanes 0..4)
};
#define SC_IO_WORDS (3 * 64 + 3 * 8 + 2)  // cur_h, cur_p, cur_k per lane; dg_h, dg_slot, dg_ck of lanes 0..4; plvl5, adjacent
typedef u32 sc_u32x2 __attribute__((ext_vector_type(2)));

__device__ __noinline__ void score_level_slow(int *s_io_v, int prev_h, u32 w_first, int dl_v,
                                                     int n_link_v, u32 lk_v, u32 slot_v,
                                                     int cov_v, int upper_v,
                                                     int curbuf_v, const u32 *links_v,
                                                     const u32 *s_links_v, int *s_deep_v,
                                                     sc_u32x2 *nodes_v) {
    // every scalar argument is wave-uniform; pin it (arguments arrive in VGPRs)
    const int dl = fa_uni(dl_v), n_link = fa_uni(n_link_v), cov = fa_uni(cov_v);
    const int upper = fa_uni(upper_v), curbuf = fa_uni(curbuf_v);
    const u32 lk = fa_uni(lk_v), slot = fa_uni(slot_v);
    const u32 *links = fa_uni(links_v);        // the level's words in HBM ...
    const u32 *s_links = fa_uni(s_links_v);    // ... or staged in LDS (then links is unused): generic pointer
    int *s_deep = fa_uni(s_deep_v);
    sc_u32x2 *nodes = fa_uni(nodes_v);
    int *s_io = fa_uni(s_io_v);
    // (these two depend on the previous position: as register arguments they would pull the
    // caller's loop-carried scalars into vector registers)
    const u32 plvl5 = (u32)fa_uni(s_io[216]);
    const bool adjacent = fa_uni(s_io[217]) != 0;
    const int lane = fa_lane();
    ScoreSlowIo io;
    io.cur_h = s_io[lane]; io.cur_p = s_io[64 + lane]; io.cur_k = s_io[128 + lane];
    io.dg_h = s_io[192 + (lane & 7)]; io.dg_slot = s_io[200 + (lane & 7)]; io.dg_ck = s_io[208 + (lane & 7)];
    const bool have = lane < n_link;
    const u32 w = w_first;
    const int cnt = (int)(w & LW_CNT_MASK), nbase = (int)((w >> LW_NB_SHIFT) & 7u);
    const int pidx = (int)((w >> LW_PIDX_SHIFT) & 0x7ffu);
    const bool start = (w >> LW_START_BIT) & 1u;
    (void)have;
    if (dl < SC_REG && n_link <= 64 &&
        // predecessors beyond the register-resident levels take the generic path
        // (start links carry pidx 0)
        (fa_ballot(pidx >= SC_REG * 5) & (n_link >= 64 ? ~0ull : ((1ull << n_link) - 1ull))) == 0ull) {
        ScoreAcc acc;
        acc.h = io.cur_h; acc.p = io.cur_p; acc.k = io.cur_k; acc.n = 0;
        const int cv = 2 * cnt - cov;
        const int lidx = start ? SC_ZERO : pidx;
        const int pidv = start ? -1 : (int)(plvl5 + (u32)pidx);
        const int lane_rel = lane - dl * 5;
        if (dl == 0) score_links<false>(acc, prev_h, nbase, cv, lidx, pidv, n_link, lane_rel);
        else score_links<true>(acc, 0, nbase, cv, lidx, pidv, n_link, lane_rel);
        s_io[lane] = acc.h; s_io[64 + lane] = acc.p; s_io[128 + lane] = acc.k;
        return;
    }
    // generic level: any number of links, predecessors and/or the level itself beyond
    // the register-resident ones; accumulators in lanes 0..4
    ScoreAcc d;
    d.h = -2; d.p = 0; d.k = 0; d.n = 0;
    const int which = (dl == 0) ? (curbuf ^ 1) : curbuf;
    for (int c0 = 0; c0 < n_link; c0 += 64) {
        u32 wc = 0;
        if (c0 + lane < n_link) {
            if (s_links) wc = s_links[c0 + lane];
            else wc = links[lk + (u32)(c0 + lane)];
        }
        wc = fa_settled(wc);
        const int n_here = min(64, n_link - c0);
        for (int l = 0; l < n_here; l++) {
            const u32 wl = (u32)__builtin_amdgcn_readlane((int)wc, l);
            const int cnt_l = (int)(wl & LW_CNT_MASK), nb_l = (int)((wl >> LW_NB_SHIFT) & 7u);
            const int pidx_l = (int)((wl >> LW_PIDX_SHIFT) & 0x7ffu);
            const bool start_l = (wl >> LW_START_BIT) & 1u;
            int ph = 0, pid = -1;
            if (!start_l) {
                if (pidx_l < SC_REG * 5) {
                    ph = (dl == 0) ? __builtin_amdgcn_readlane(prev_h, pidx_l)
                                   : __builtin_amdgcn_readlane(io.cur_h, pidx_l);
                } else {
                    ph = __builtin_amdgcn_readfirstlane(s_deep[which * 1280 + pidx_l]);
                    if (dl == 0 && !adjacent) ph = -2;
                }
                pid = (int)(plvl5 + (u32)pidx_l);
            }
            const int h = ph + 2 * cnt_l - cov;
            const bool mine = lane == nb_l;
            const bool better = mine && h > d.h;
            d.h = better ? h : d.h;
            d.p = better ? pid : d.p;
            d.k = better ? d.n : d.k;
            d.n += mine ? 1 : 0;
        }
    }
    if (dl < SC_REG) {  // hand the five nodes to their score lanes
        const int src = lane - dl * 5;
        const int vh = __shfl(d.h, src), vp = __shfl(d.p, src), vk = __shfl(d.k, src);
        if (src >= 0 && src < 5) { io.cur_h = vh; io.cur_p = vp; io.cur_k = vk; }
    } else if (lane < 5) {
        s_deep[curbuf * 1280 + dl * 5 + lane] = d.h;
        __threadfence_block();
        sc_u32x2 r;
        r.x = (u32)d.h; r.y = (u32)(((d.p + 1) << 1) | upper);
        nodes[slot * 5u + (u32)lane] = r;
        if (d.h > io.dg_h) { s_io[192 + lane] = d.h; s_io[200 + lane] = (int)slot; s_io[208 + lane] = d.k; }
    }
    s_io[lane] = io.cur_h; s_io[64 + lane] = io.cur_p; s_io[128 + lane] = io.cur_k;
}

__global__ __launch_bounds__(64) void k_score1(MsaArgs A) {
    __shared__ int s_io[SC_IO_WORDS];  // score_level_slow's state (and the deep levels' best nodes)
    const int lane = fa_lane();
    const int p = blockIdx.x;
    if (p >= A.n_pile) return;
    const FaPile pm = A.pile[p];
    FaScoreOut so = A.score_out[p];
    // (this kernel also writes score_out, so the record arrives by vector loads: pin what
    // steers the control flow, or every branch on it counts as divergent and the values
    // merged behind it move to vector registers)
    so.err = fa_uni(so.err); so.wide = fa_uni(so.wide);
    so.n_levels = fa_uni(so.n_levels); so.n_links = fa_uni(so.n_links);
    if (so.err) return;
    if (A.only_redo && fa_uni(so.redo) == 0) return;
    const int T = pm.seed_len;
    const u32 *tiw = reinterpret_cast<const u32 *>(A.tinfo + A.t_off[p]);  // 3 words per position
    const u32 *links = A.links + A.link_off[p];
    const u16 *nlk = A.lvl_nlink16 + pm.node_off / 5;
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    u32x2 *nodes = reinterpret_cast<u32x2 *>(A.nodes + pm.node_off);
    const int min_cov = (int)A.min_cov;
    // scores of levels >= SC_REG (rare): previous / current position, 2 x 256 x 5 ints in
    // HBM -- LDS here would cost occupancy (one wave per pile must all be resident)
    int *s_deep = A.score_ovf + (u64)p * (2 * 256 * 5);
    const int ldl = lane / 5;
    const int h_init = (lane == SC_ZERO) ? 0 : -2;
    (void)A.force_generic;  // (every level takes the general path now)

    ScoreAcc cur;
    cur.h = h_init; cur.p = 0; cur.k = 0; cur.n = 0;
    int prev_h = h_init;
    int gl_h = -2, gl_slot = 0, gl_ck = 0;  // best node of this lane's (delta, base) class
    // (the same for the deep levels, lanes 0..4: s_io[192..], [200..], [208..])
    if (lane < 8) { s_io[192 + lane] = -2; s_io[200 + lane] = 0; s_io[208 + lane] = 0; }
    int curbuf = 0;     // which half of s_deep belongs to the position being scored
    int prev_nlev = 0;  // its number of levels
    int prev_t = -2;    // last scored target position
    u32 prev_lvl = 0;   // its first level slot
    u32 lk_run = 0;     // where the next position's links start: k_links writes the links of a segment of
                        // TSEG positions back to back from the segment's first link slot

    if (lane < 5 && (tiw[2] & 0xffffu) == 0u) {
        // slot 0 = (t 0, delta 0): absent nodes, target of the zero back pointer (Q4)
        u32x2 r;
        r.x = (u32)-2; r.y = (u32)((0 + 1) << 1);
        nodes[lane] = r;
    }
    int t0 = 0;
    while (t0 < T) {
        // ---- a block of positions: as many (<= 63) as fit the staging budgets.
        // Lane j holds position t0+j; a block of length j ends where position t0+j
        // starts (or at the pile totals), so lane j can judge whether length j fits.
        const int tl = t0 + lane;
        u32 x_lvl = 0, x_link = 0, x_cn = 0;
        if (tl < T) { x_lvl = tiw[3 * tl]; x_link = tiw[3 * tl + 1]; x_cn = tiw[3 * tl + 2]; }
        const u32 lvl0 = (u32)__builtin_amdgcn_readfirstlane((int)x_lvl);
        const u32 lnk0 = (u32)__builtin_amdgcn_readfirstlane((int)x_link);
        // (ONE position at a time, its link words read straight from HBM: the position records
        // carry a link slot per SEGMENT only, so a block of positions cannot be bounded in the
        // link words and staged -- rounds 1-3 did that; this is the kernel behind k_score2, for
        // the handful of piles that one hands on and for the unitig call)
        (void)lvl0;
        const int nb = 1;
        if ((t0 & (TSEG - 1)) == 0) lk_run = lnk0;
        lk_run = fa_uni(lk_run);
        x_lvl = fa_settled(x_lvl);
        x_link = fa_settled(x_link);
        x_cn = fa_settled(x_cn);

        // (what the position loop carries from position to position is wave-uniform; pinned once
        // per block, the loop keeps it in scalar registers instead of following the vector
        // copies the other instance's joins hand it)
        prev_t = fa_uni(prev_t);
        prev_lvl = fa_uni(prev_lvl);
        prev_nlev = fa_uni(prev_nlev);
        curbuf = fa_uni(curbuf);
        {
        // (covered positions only: a `continue` for the others is a second way to the loop
        // latch and costs every position a round of register copies)
        for (u64 todo = fa_ballot((x_cn & 0xffffu) != 0u) & ((1ull << nb) - 1ull); todo; todo &= todo - 1) {
            const int j = (int)__builtin_ctzll(todo);
            const int t = t0 + j;
            const u32 y_lvl = (u32)__builtin_amdgcn_readlane((int)x_lvl, j);
            const u32 y_cn = (u32)__builtin_amdgcn_readlane((int)x_cn, j);
            const int cov = (int)(y_cn & 0xffffu), nlev = (int)(y_cn >> 16);
            // falcon.c:498 (Q7): cov > min_cov -- as scalar integer arithmetic the compiler cannot
            // turn back into a compare: a uniform bool lives as a lane mask, and its `? 1 : 0`
            // is a vector select in every position, wanted or not
            int upper;
            asm("s_sub_i32 %0, %1, %2\n\ts_lshr_b32 %0, %0, 31" : "=s"(upper) : "s"(min_cov), "s"(cov) : "scc");
            const bool adjacent = (prev_t == t - 1);
            int adjacent_i;  // (the same, as an integer: see upper; t - 1 - prev_t >= 0)
            asm("s_sub_i32 %0, %1, %2\n\ts_min_u32 %0, %0, 1\n\ts_xor_b32 %0, %0, 1"
                : "=s"(adjacent_i) : "s"(t - 1), "s"(prev_t) : "scc");
            prev_h = adjacent ? cur.h : h_init;
            cur.h = h_init; cur.p = 0; cur.k = 0;
            curbuf ^= 1;
            u32 lk = lk_run;
            for (int dl = 0; dl < nlev; dl++) {
                const u32 slot = y_lvl + (u32)dl;
                const int n_link = __builtin_amdgcn_readfirstlane((int)nlk[slot]);
                const u32 plvl5 = (dl == 0 ? prev_lvl : y_lvl) * 5u;  // node id = plvl5 + pidx
                // first 64 links of the level, lanes = links
                u32 w = 0;
                if (lane < n_link) w = links[lk + (u32)lane];
                w = fa_settled(w);
                {
                    s_io[lane] = cur.h; s_io[64 + lane] = cur.p; s_io[128 + lane] = cur.k;
                    // (scalar -> vector here and nowhere else: the "s" operands keep the
                    // loop-carried scalars they derive from in scalar registers)
                    int plvl5_v, adj_v, curbuf_v, upper_v, cov_v;
                    asm volatile("v_mov_b32 %0, %5\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %7\n\t"
                                 "v_mov_b32 %3, %8\n\tv_mov_b32 %4, %9"
                                 : "=&v"(plvl5_v), "=&v"(adj_v), "=&v"(curbuf_v), "=&v"(upper_v), "=&v"(cov_v)
                                 : "s"(plvl5), "s"(adjacent_i), "s"(curbuf), "s"(upper), "s"(cov));
                    s_io[216] = plvl5_v; s_io[217] = adj_v;
                    const u32 *staged = nullptr;  // (no staged copy of the links: see above)
                    score_level_slow(s_io, prev_h, w, dl, n_link, lk, slot, cov_v, upper_v,
                                     curbuf_v, links, staged, s_deep, (sc_u32x2 *)nodes);
                    cur.h = s_io[lane]; cur.p = s_io[64 + lane]; cur.k = s_io[128 + lane];
                }
                lk += (u32)n_link;
            }
            // the register-resident levels of the position: node records + lane bests
            {
                const bool in = lane < min(nlev, SC_REG) * 5;
                if (in) {
                    u32x2 r;
                    r.x = (u32)cur.h; r.y = (u32)(((cur.p + 1) << 1) | upper);
                    nodes[y_lvl * 5u + (u32)lane] = r;
                }
                // strict: the lane's first maximum (selects, in place: a masked block costs
                // copies in and out of it)
                const bool better = in && cur.h > gl_h;
                gl_h = better ? cur.h : gl_h;
                gl_slot = better ? (int)y_lvl + ldl : gl_slot;
                gl_ck = better ? cur.k : gl_ck;
            }
            // (defined by scalar instructions, so that the loop keeps them in scalar registers)
            asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=&s"(prev_t), "=&s"(prev_lvl) : "s"(t), "s"(y_lvl));
            prev_nlev = nlev;
            lk_run = lk;
        }
        }
        t0 += nb;
    }
    // global best = first strict maximum in (t, delta, base) order (falcon.c:464-469):
    // the highest score; among equals the lowest level slot, then the lowest base
    int g_h = -2, g_node = -1, g_ck = 0, g_slot = 0x7fffffff;
    for (int i = 0; i < SC_REG * 5 + 5; i++) {
        const bool dp = i >= SC_REG * 5;
        const int src = dp ? i - SC_REG * 5 : i;
        const int hv = dp ? fa_uni(s_io[192 + src]) : __builtin_amdgcn_readlane(gl_h, src);
        const int sv = dp ? fa_uni(s_io[200 + src]) : __builtin_amdgcn_readlane(gl_slot, src);
        const int kv = dp ? fa_uni(s_io[208 + src]) : __builtin_amdgcn_readlane(gl_ck, src);
        if (hv > g_h || (hv == g_h && hv > -2 && sv < g_slot)) {
            g_h = hv;
            g_slot = sv;
            g_node = sv * 5 + src % 5;
            g_ck = kv;
        }
    }
    so.g_h = g_h;
    so.g_node = g_node;
    so.g_ck = g_ck;
    A.score_out[p] = so;  // every lane stores the same record
}

void fa_launch_score1(const MsaArgs &A, hipStream_t s) {
    hipLaunchKernelGGL(k_score1, dim3(A.n_pile), dim3(64), 0, s, A);
}

// (fa_warm: the code object of this file is loaded when one of its kernels is first looked at)
void fa_touch_score1() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_score1));
}
