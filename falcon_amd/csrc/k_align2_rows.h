// k_align2_rows.h -- the row loop of k_align2 as ONE hand-scheduled gfx950 instruction stream.
//
// What it computes is stated by a2_rows_c (k_align2_core.h): rows of up to two tracks -- the
// furthest-reaching points of DW_banded.c:183-243 -- from h.it on, until a row raises an event
// (a cell reached an end of a sequence, a band hull reached a forbidden lane, a snake does not
// fit its cell byte) or iteration `it_end` is reached; the cell bytes of every fourth and the
// records of every 64th iteration go to the tape on the way.  The lane emulator (tests/emu)
// runs a2_rows_c; the device runs this.  Why by hand: k_align2 is bound by instruction issue,
// and the compiler's rendering of a2_rows_c was 48 vector + 50 scalar instructions per
// iteration (profiles/r04_pmc_table.txt) -- half of the scalar ones the state machine it made
// of the loop's exits, the hull algebra and the bookkeeping.  This one is 39 + 2 memory + 24:
//
//   * x = max(V[k-1] + 1, V[k+1]): the lanes outside a band's hull hold A2_NEG, so the maximum IS
//     the reference's choice (from above at min_k, from below at max_k) and the from_above bit
//     is only needed for the tape.  The lane below lane 0 / above lane 63 is A2_NEG too: the
//     DPP shifts leave lane 0 of `tdn` / lane 63 of `tup` alone, and those hold A2_NEG for good.
//   * per-lane constants fold the diagonal: query address x + cq, target address x + ct, the
//     bases left min(q_len - x, t_len - y) = L - x, the key x + y = 2 x + ck (L and ck one
//     register per row parity).  A cell has reached an end of a sequence iff x >= L.
//   * keys are signed: a lane outside the bands computes x = A2_NEG + 1 and a negative key all
//     by itself -- no masks on the vector side; the loads alone run under exec = act.
//   * best_m by one prefix maximum for both tracks (track 1's keys carry A2_TOP), the filter as
//     two compares against the two scalars.
//   * the hull of the lanes that pass the filter: 99 % of the rows they are contiguous per track
//     (measured on the lane emulator) -- then the hull IS the mask; the test costs five scalar
//     instructions, the general hull algebra (out of line) thirteen.
//   * the cells of a track are counted per LANE (one v_add under exec = act), summed when the
//     lanes change hands (a2_fold_cells).
//   * the snake beyond 16 bases (a fifth of the rows), snakes of >= 255 bases, the tape stores:
//     out of line.  Two `act` registers take turns, so a row never copies a mask.
//   * exits: one label.  Which row left and why is worked out by the caller from what the
//     stream hands back (the row's masks, the iteration counter).
//
// Round 6: (1) the PAIR stream lays its two bands out again ITSELF (.La2r_relay: what a2_replace<true> states --
// hulls, the free lanes shared out afresh, the new boundary, ds_bpermute of the row, the tape's K fields)
// and computes the per-lane constants of both tracks from scalars (.La2r_const), so a band hull on a fence
// lane no longer leaves the asm statement: 38 M such events per launch were a round trip of ~250 instructions
// each, a quarter of the kernel.  What it declines (a cell at an end of a sequence, a long snake, bands that no
// longer fit together, a row 0) leaves through the old exits.  (2) The instructions are chosen by what they
// cost the vector pipe (profiles/r06_ubench_valu_cost.txt: v_add / v_sub / v_and / v_or / v_xor / v_mov /
// v_lshrrev 2 cycles per SIMD, everything else -- v_lshlrev, v_max, compares, DPP, three-operand forms,
// ANY instruction with an SGPR operand -- 3.6): x + x for the left shifts, VGPR copies of `band` and of
// the odd rows' target constant.  (3) cells are counted per lane in two 16-bit halves (track 0 / track 1:
// the increment is a per-lane constant), so lanes that change hands need no fold.
//
// Hazards kept by hand (the assembler does not insert wait states): VALU write -> DPP read of
// the same VGPR: 2 wait states; VALU write -> v_readlane of it: 1; VALU writes an SGPR -> use
// as a lane select: none here (selects come from the scalar unit).
#pragma once

// ---- the first 16 bases of a snake: x in %[x], leaves min(matching bases, bases left, 16) in DST
// TA: the instruction that forms the target address, LIM: the register with L of this row parity
// QA / TA: the instructions that form the query / target address (bases: LDS window or global),
// LQ / LT: the two loads, W1 / W0: the waits for the first / both, LIM: the register with L of this
// row parity, MID: what runs in the shadow of the loads
#define A2R_SNAKE16(...) A2R_SNAKE16_(__VA_ARGS__)
#define A2R_SNAKE16_(QA, TA, LQ, LT, W1, W0, LIM, MID)             \
    QA                                                             \
    TA                                                             \
    "v_lshrrev_b32 %[t1], 2, %[qa]\n\t"                            \
    "v_lshrrev_b32 %[t2], 2, %[ta]\n\t"                            \
    "v_and_b32 %[t1], -4, %[t1]\n\t"                               \
    "v_and_b32 %[t2], -4, %[t2]\n\t"                               \
    LQ                                                             \
    LT                                                             \
    MID                                                            \
    "v_add_u32 %[qa], %[qa], %[qa]\n\t"                            \
    "v_add_u32 %[ta], %[ta], %[ta]\n\t"                            \
    "v_sub_u32 %[t1], " LIM ", %[x]\n\t"                           \
    W1                                                             \
    "v_alignbit_b32 v52, v53, v52, %[qa]\n\t"                      \
    W0                                                             \
    "v_alignbit_b32 v54, v55, v54, %[ta]\n\t"                      \
    "v_xor_b32 v52, v52, v54\n\t"                                  \
    "v_ffbl_b32 v52, v52\n\t"                                      \
    "v_lshrrev_b32 v52, 1, v52\n\t"

// the rows read the two sequences out of the wavefront's LDS windows (a2_win_fill) ...
#define A2R_QA_LDS "v_add_u32 %[qa], %[x], %[cq]\n\t"
#define A2R_TA_EVEN "v_add_u32 %[ta], %[x], %[ct]\n\t"
#define A2R_TA_ODD "v_add3_u32 %[ta], %[x], %[ct], -1\n\t"
#define A2R_LDS_LOADS "ds_read2_b32 v[52:53], %[t1] offset1:1\n\t", "ds_read2_b32 v[54:55], %[t2] offset1:1\n\t", \
                      "s_waitcnt lgkmcnt(1)\n\t", "s_waitcnt lgkmcnt(0)\n\t"
// ... a snake beyond its first 16 bases goes on in global memory (it may leave any window)
#define A2R_QA_GLB "v_add_u32 %[qa], %[x], %[cqg]\n\t"
#define A2R_TA_EVEN_GLB "v_add_u32 %[ta], %[x], %[ctg]\n\t"
#define A2R_TA_ODD_GLB "v_add3_u32 %[ta], %[x], %[ctg], -1\n\t"   /* (the far path: a fifth of the rows) */
#define A2R_GLB_LOADS "global_load_dwordx2 v[52:53], %[t1], %[words]\n\t", "global_load_dwordx2 v[54:55], %[t2], %[words]\n\t", \
                      "s_waitcnt vmcnt(1)\n\t", "s_waitcnt vmcnt(0)\n\t"

// ---- best_m and the lanes that pass the band filter, into %[in]
#define A2R_BEST_PAIR                                              \
    "v_readlane_b32 %[p0], %[pm], %[ln0]\n\t"                      \
    "v_readlane_b32 %[p1], %[pm], 63\n\t"                          \
    "s_max_i32 %[b0], %[b0], %[p0]\n\t"                            \
    "s_max_i32 %[b1], %[b1], %[p1]\n\t"                            \
    "s_sub_i32 %[p0], %[b0], %[band]\n\t"                          \
    "s_sub_i32 %[p1], %[b1], %[band]\n\t"                          \
    "v_cmp_le_i32 vcc, %[p0], %[key]\n\t"                          \
    "v_cmp_le_i32 %[c1], %[p1], %[key]\n\t"                        \
    "s_andn2_b64 %[in], vcc, %[z1]\n\t"                            \
    "s_or_b64 %[in], %[in], %[c1]\n\t"
#define A2R_BEST_SINGLE                                            \
    "v_readlane_b32 %[p1], %[pm], 63\n\t"                          \
    "s_max_i32 %[b0], %[b0], %[p1]\n\t"                            \
    "s_sub_i32 %[p1], %[b0], %[band]\n\t"                          \
    "v_cmp_le_i32 %[in], %[p1], %[key]\n\t"

// ---- the general hull of %[in] into %[c1] (out of line)
#define A2R_HULL_PAIR                                              \
    "s_andn2_b64 %[c1], %[in], %[z1]\n\t"                          \
    "s_ff1_i32_b64 %[p0], %[c1]\n\t"                               \
    "s_flbit_i32_b64 %[p1], %[c1]\n\t"                             \
    "s_lshl_b64 %[t], -1, %[p0]\n\t"                               \
    "s_lshr_b64 %[c1], -1, %[p1]\n\t"                              \
    "s_and_b64 %[c1], %[c1], %[t]\n\t"                             \
    "s_and_b64 %[t], %[in], %[z1]\n\t"                             \
    "s_ff1_i32_b64 %[p0], %[t]\n\t"                                \
    "s_flbit_i32_b64 %[p1], %[t]\n\t"                              \
    "s_lshl_b64 %[t], -1, %[p0]\n\t"                               \
    "s_lshr_b64 %[u], -1, %[p1]\n\t"                               \
    "s_and_b64 %[t], %[t], %[u]\n\t"                               \
    "s_or_b64 %[c1], %[c1], %[t]\n\t"
#define A2R_HULL_SINGLE                                            \
    "s_ff1_i32_b64 %[p0], %[in]\n\t"                               \
    "s_flbit_i32_b64 %[p1], %[in]\n\t"                             \
    "s_lshl_b64 %[t], -1, %[p0]\n\t"                               \
    "s_lshr_b64 %[c1], -1, %[p1]\n\t"                              \
    "s_and_b64 %[c1], %[c1], %[t]\n\t"

// ---- one row.  J: the iteration's byte in the cell word (= it & 3), its parity the row's phase.
//   DPP:   the shift of the previous row into tdn / tup      A1: V[k-1] + 1 into %[x]
//   FA:    from_above into vcc                               MX: x = max(V[k-1] + 1, V[k+1])
//   TA / LIM / CK: target address, L and key constant of this parity
//   RD / WR: the act register the row reads / writes         SH: s_lshr_b64 (even) / s_lshl_b64 (odd)
//   FORB:  the lanes the hull may not reach in this row      SEL: the v_perm selector of byte J
//   F1 / F2: what fills the wait states of the second and third DPP step (two each)
//   BEST / HULL / NRUN: pair or single
#define A2R_ROW(J, DPP, A1, FA, MX, TA, LIM, LW, CK, RD, WR, SH, FORB, OUT, EVT, SEL, F1, F2, BEST, HULL, NRUN) \
    ".La2r_r" J "_%=:\n\t"                                                                               \
    DPP                                                                                                  \
    A1                                                                                                   \
    FA                                                                                                   \
    MX                                                                                                   \
    "s_mov_b64 exec, " RD "\n\t"                                                                         \
    "v_add_u32 %[cnt], %[cinc], %[cnt]\n\t"                                                              \
    A2R_SNAKE16(A2R_QA_LDS, TA, A2R_LDS_LOADS, LIM,                                                      \
                "v_writelane_b32 %[mlo], vcc_lo, m0\n\t"                                                 \
                "v_writelane_b32 %[mhi], vcc_hi, m0\n\t")                                                \
    "v_min3_u32 %[vm], v52, %[t1], 16\n\t"                                                               \
    "v_cmp_eq_u32 vcc, 16, %[vm]\n\t"                                                                    \
    "v_add_u32 %[x], %[x], %[vm]\n\t"                                                                    \
    "s_cbranch_vccnz .La2r_x" J "_%=\n"                                                                  \
    ".La2r_b" J "_%=:\n\t"                                                                               \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "v_lshl_add_u32 %[key], %[x], 1, " CK "\n\t"                                                         \
    "v_perm_b32 %[acc], %[vm], %[acc], " SEL "\n\t"                                                      \
    "s_add_u32 m0, m0, 1\n\t"                                                                            \
    "v_max_i32_dpp %[pm], %[key], %[key] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"          \
    "v_cmp_ge_i32 %[fin], %[x], " LW "\n\t"                                                              \
    "v_max_i32_dpp %[pm], %[pm], %[pm] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                         \
    F1                                                                                                   \
    "v_max_i32_dpp %[pm], %[pm], %[pm] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                         \
    F2                                                                                                   \
    "v_max_i32_dpp %[pm], %[pm], %[pm] row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                         \
    "s_nop 1\n\t"                                                                                        \
    "v_max_i32_dpp %[pm], %[pm], %[pm] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                      \
    "s_nop 1\n\t"                                                                                        \
    "v_max_i32_dpp %[pm], %[pm], %[pm] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                      \
    "s_nop 1\n\t"                                                                                        \
    BEST                                                                                                 \
    SH " %[t], %[in], 1\n\t"                                                                             \
    "s_andn2_b64 %[c1], %[in], %[t]\n\t"                                                                 \
    "s_bcnt1_i32_b64 %[p0], %[c1]\n\t"                                                                   \
    "s_cmp_lg_u32 %[p0], " NRUN "\n\t"                                                                   \
    "s_cbranch_scc1 .La2r_s" J "_%=\n\t"                                                                 \
    "v_cndmask_b32 %[vx], %[neg], %[x], %[in]\n\t"                                                       \
    "s_or_b64 " WR ", %[in], %[t]\n"                                                                     \
    ".La2r_h" J "_%=:\n\t"                                                                               \
    "s_and_b64 %[t], %[in], " FORB "\n\t"                                                                \
    "s_or_b64 %[t], %[t], %[fin]\n\t"                                                                    \
    "s_cbranch_scc1 " EVT J "_%=\n"                                                                       \
    ".La2r_c" J "_%=:\n\t"                                                                               \
    "s_sub_u32 %[rem], %[rem], 1\n\t"                                                                    \
    "s_cbranch_scc1 " OUT "\n\t"

// ---- what a row keeps out of line: the snake beyond 16 bases (then: snakes of >= 255 bases end
// the stretch with this row), the hull of a filter mask with holes
#define A2R_ROW_FAR(J, TAG, LIM, RD, WR, SH, HULL)                                                       \
    ".La2r_x" J "_%=:\n\t"                                                                               \
    "s_mov_b64 %[t], vcc\n"                                                                              \
    ".La2r_xl" J "_%=:\n\t"                                                                              \
    "s_mov_b64 exec, %[t]\n\t"                                                                           \
    A2R_SNAKE16(A2R_QA_GLB, TAG, A2R_GLB_LOADS, LIM, "")                                                 \
    "v_min3_u32 v52, v52, %[t1], 16\n\t"                                                                 \
    "v_add_u32 %[x], %[x], v52\n\t"                                                                      \
    "v_add_u32 %[vm], %[vm], v52\n\t"                                                                    \
    "v_cmp_eq_u32 vcc, 16, v52\n\t"                                                                      \
    "s_and_b64 %[t], vcc, exec\n\t"                                                                      \
    "s_cbranch_scc1 .La2r_xl" J "_%=\n\t"                                                                \
    "s_mov_b64 exec, " RD "\n\t"                                                                         \
    "v_cmp_lt_u32 vcc, 0xfe, %[vm]\n\t"                                                                  \
    "s_mov_b64 %[big], vcc\n\t"                                                                          \
    "s_cmp_eq_u64 %[big], 0\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_b" J "_%=\n\t"                                                                 \
    "s_mov_b32 %[rem], 0\n\t"                                                                            \
    "s_branch .La2r_b" J "_%=\n"                                                                         \
    ".La2r_s" J "_%=:\n\t"                                                                               \
    HULL                                                                                                 \
    "v_cndmask_b32 %[vx], %[neg], %[x], %[c1]\n\t"                                                       \
    SH " %[t], %[c1], 1\n\t"                                                                             \
    "s_or_b64 " WR ", %[c1], %[t]\n\t"                                                                   \
    "s_branch .La2r_h" J "_%=\n"

#define A2R_EVEN(J, EVT, SEL, F1, F2, BEST, HULL, NRUN)                                                  \
    A2R_ROW(J, "v_mov_b32_dpp %[tdn], %[vx] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t",                  \
            "v_add_u32 %[x], 1, %[tdn]\n\t",                                                             \
            "v_cmp_lt_i32 vcc, %[tdn], %[vx]\n\t",                                                       \
            "v_max_i32 %[x], %[x], %[vx]\n\t",                                                           \
            A2R_TA_EVEN, "%[le]", "%[lwe]", "%[cke]", "%[sa]", "%[sb]", "s_lshr_b64", "%[f1]", ".La2r_lk" J "_%=", EVT, SEL, F1, F2, BEST, HULL, NRUN)
#define A2R_ODD(J, EVT, SEL, F1, F2, BEST, HULL, NRUN)                                                   \
    A2R_ROW(J, "v_mov_b32_dpp %[tup], %[vx] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t",                  \
            "v_add_u32 %[x], 1, %[vx]\n\t",                                                              \
            "v_cmp_lt_i32 vcc, %[vx], %[tup]\n\t",                                                       \
            "v_max_i32 %[x], %[x], %[tup]\n\t",                                                          \
            A2R_TA_ODD, "%[lo]", "%[lwo]", "%[cko]", "%[sb]", "%[sa]", "s_lshl_b64", "%[f0]", ".La2r_lk" J "_%=", EVT, SEL, F1, F2, BEST, HULL, NRUN)
#define A2R_EVEN_FAR(J, HULL) A2R_ROW_FAR(J, A2R_TA_EVEN_GLB, "%[le]", "%[sa]", "%[sb]", "s_lshr_b64", HULL)
#define A2R_ODD_FAR(J, HULL) A2R_ROW_FAR(J, A2R_TA_ODD_GLB, "%[lo]", "%[sb]", "%[sa]", "s_lshl_b64", HULL)

#define A2R_NOP2 "s_nop 1\n\t"
// the fourth row of a group: its two fillers send the lanes' cell words to the tape and look
// whether the block of 64 iterations is complete (m0 counts the iteration already)
#define A2R_F1_STORE                                                                                     \
    "v_add_u32 %[t2], %[coff], %[l4]\n\t"                                                                \
    "s_add_u32 %[coff], %[coff], 0x100\n\t"
#define A2R_F2_STORE                                                                                     \
    "global_store_dword %[t2], %[acc], %[cells]\n\t"                                                     \
    "s_and_b32 %[coff], %[coff], %[cmaskb]\n\t"                                                          \
    "s_cmp_eq_u32 m0, 64\n\t"                                                                            \
    "s_cbranch_scc1 .La2r_rec_%=\n"                                                                      \
    ".La2r_recb_%=:\n\t"

// the records of a complete block of 64 iterations to the tape (one 16-byte store per lane through the
// snake's four registers, free at this point of a row: four dword stores 16 bytes apart cost 12 GB of
// partial-line writes per launch, profiles/r05_pmc_traffic*); the K fields start over
#define A2R_REC                                                                                          \
    ".La2r_rec_%=:\n\t"                                                                                  \
    "v_lshl_add_u32 %[t2], %[l4], 2, %[roff]\n\t"                                                        \
    "v_mov_b32 v52, %[mlo]\n\t"                                                                          \
    "v_mov_b32 v53, %[mhi]\n\t"                                                                          \
    "v_mov_b32 v54, %[k0]\n\t"                                                                           \
    "v_mov_b32 v55, %[k1]\n\t"                                                                           \
    "s_nop 0\n\t"                                                                                        \
    "global_store_dwordx4 %[t2], v[52:55], %[recs]\n\t"                                                  \
    "s_nop 1\n\t"                                                                                        \
    "s_add_u32 %[roff], %[roff], 0x400\n\t"                                                              \
    "s_and_b32 %[roff], %[roff], %[rmaskb]\n\t"                                                          \
    "s_add_u32 %[itb], %[itb], 64\n\t"                                                                   \
    "s_mov_b32 m0, 0\n\t"                                                                                \
    "v_mov_b32 %[k0], %[kb0]\n\t"                                                                        \
    "v_mov_b32 %[k1], %[kb1]\n\t"                                                                        \
    "s_branch .La2r_recb_%=\n"

// ---- PAIR only: the per-lane constants of both tracks from the tracks' scalars.  Lane l of track t holds diagonal
// kb_t - 1 + 2 l before an even row: vnegk = 1 - kb_t - 2 l; with the track's scalars
//   cq = CQ_t                 ct = CT_t + vnegk          cqg = QB_t        ctg = TB_t + vnegk
//   le = min(QL_t, TL_t - vnegk)    lo = min(QL_t, TL_t - vnegk + 1)     (a cell is at an end iff x >= L)
//   lw* = min(l*, min(ROOMQ_t - cq, ROOMT_t - ct))                       (... or within the margin of a window)
//   cke = vnegk + top_t       cko = cke - 1              cinc = 1 / 0x10000 (the lane's half of the cell counter)
// The scalars live in the lanes of `sc` (a2_fast's packed wave-uniform state, A2_SC_QLEN0 .. A2_SC_CT1: sixteen
// more SGPR operands do not fit the function's budget) and reach all lanes by ds_bpermute with a zero address and
// the lane's byte offset in the instruction (no LDS access, no vector instruction): twelve at once into registers
// that are dead at this point -- the row's temporaries and the constants about to be made again -- then one select
// by zone each and arithmetic on VGPR operands.  A re-layout issues them BEFORE its scalar algebra, so they have
// landed when it is done.  `ret` says where to go on: 0..3 = behind row J's event test (then the moved row, which
// ds_bpermute left in `cinc`, goes into vx here), else the stretch's first row.
#define A2R_STR_(x) #x
#define A2R_STR(x) A2R_STR_(x)
#define A2R_BC(DST, IDX) "ds_bpermute_b32 " DST ", %[t1], %[sc] offset:" A2R_STR(IDX) "\n\t"
#define A2R_BCAST12                                                                                      \
    "v_mov_b32 %[t1], 0\n\t"                                                                             \
    A2R_BC("v52", A2R_O_QL0) A2R_BC("v53", A2R_O_TL0) A2R_BC("v54", A2R_O_QB0) A2R_BC("v55", A2R_O_TB0)  \
    A2R_BC("%[pm]", A2R_O_CQ0) A2R_BC("%[vm]", A2R_O_CT0)                                                \
    A2R_BC("%[qa]", A2R_O_QL1) A2R_BC("%[ta]", A2R_O_TL1) A2R_BC("%[key]", A2R_O_QB1)                    \
    A2R_BC("%[lwe]", A2R_O_TB1) A2R_BC("%[lwo]", A2R_O_CQ1) A2R_BC("%[cko]", A2R_O_CT1)
#define A2R_CONST_PAIR                                                                                   \
    A2R_BCAST12                                                                                          \
    "s_waitcnt lgkmcnt(0)\n"                                                                             \
    ".La2r_const_%=:\n\t"                                                                                \
    "v_lshrrev_b32 %[t1], 1, %[l4]\n\t"                                                                  \
    "s_sub_u32 %[p0], 1, %[kb0]\n\t"                                                                     \
    "s_sub_u32 %[p1], 1, %[kb1]\n\t"                                                                     \
    "v_sub_u32 %[t2], %[p0], %[t1]\n\t"                                                                  \
    "s_mov_b64 exec, %[z1]\n\t"                                                                          \
    "v_sub_u32 %[t2], %[p1], %[t1]\n\t"                                                                  \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "v_cndmask_b32 %[cq], %[pm], %[lwo], %[z1]\n\t"                                                      \
    "v_cndmask_b32 %[ct], %[vm], %[cko], %[z1]\n\t"                                                      \
    "v_cndmask_b32 %[cqg], v54, %[key], %[z1]\n\t"                                                       \
    "v_cndmask_b32 %[ctg], v55, %[lwe], %[z1]\n\t"                                                       \
    "v_cndmask_b32 v53, v53, %[ta], %[z1]\n\t"                                                           \
    "v_cndmask_b32 v52, v52, %[qa], %[z1]\n\t"                                                           \
    "v_add_u32 %[ct], %[ct], %[t2]\n\t"                                                                  \
    "v_add_u32 %[ctg], %[ctg], %[t2]\n\t"                                                                \
    "v_sub_u32 %[qa], v53, %[t2]\n\t"                                                                    \
    "v_min_i32 %[le], v52, %[qa]\n\t"                                                                    \
    "v_add_u32 %[ta], 1, %[qa]\n\t"                                                                      \
    "v_min_i32 %[lo], v52, %[ta]\n\t"                                                                    \
    "v_sub_u32 %[key], " A2R_STR(A2R_ROOMQ0) ", %[cq]\n\t"                                               \
    "v_sub_u32 %[pm], " A2R_STR(A2R_ROOMT0) ", %[ct]\n\t"                                                \
    "v_mov_b32 %[cke], %[t2]\n\t"                                                                        \
    "s_mov_b64 exec, %[z1]\n\t"                                                                          \
    "v_sub_u32 %[key], " A2R_STR(A2R_ROOMQ1) ", %[cq]\n\t"                                               \
    "v_sub_u32 %[pm], " A2R_STR(A2R_ROOMT1) ", %[ct]\n\t"                                                \
    "v_add_u32 %[cke], 0x20000000, %[t2]\n\t"                                                            \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "v_min_i32 %[key], %[key], %[pm]\n\t"                                                                \
    "v_min_i32 %[lwe], %[le], %[key]\n\t"                                                                \
    "v_min_i32 %[lwo], %[lo], %[key]\n\t"                                                                \
    "v_add_u32 %[cko], -1, %[cke]\n\t"                                                                   \
    "s_cmp_gt_u32 %[ret], 3\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_cinc_%=\n\t"                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                           \
    "v_cndmask_b32 %[vx], %[neg], %[cinc], %[in]\n"                                                      \
    ".La2r_cinc_%=:\n\t"                                                                                 \
    "v_mov_b32 %[cinc], 1\n\t"                                                                           \
    "s_mov_b64 exec, %[z1]\n\t"                                                                          \
    "v_mov_b32 %[cinc], 0x10000\n\t"                                                                     \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "s_cmp_eq_u32 %[ret], 0\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_c0_%=\n\t"                                                                     \
    "s_cmp_eq_u32 %[ret], 1\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_c1_%=\n\t"                                                                     \
    "s_cmp_eq_u32 %[ret], 2\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_c2_%=\n\t"                                                                     \
    "s_cmp_eq_u32 %[ret], 3\n\t"                                                                         \
    "s_cbranch_scc1 .La2r_c3_%=\n\t"
// the x from which on a window's end is nearer than the margin, as base indices into the four windows
#define A2R_ROOMQ0 3904
#define A2R_ROOMT0 8000
#define A2R_ROOMQ1 12096
#define A2R_ROOMT1 16192

// byte offsets of the tracks' scalars in `sc` (4 x A2_SC_*: the preprocessor cannot see an enum)
#define A2R_O_QL0 120
#define A2R_O_TL0 124
#define A2R_O_QL1 128
#define A2R_O_TL1 132
#define A2R_O_QB0 136
#define A2R_O_TB0 140
#define A2R_O_QB1 144
#define A2R_O_TB1 148
#define A2R_O_CQ0 236
#define A2R_O_CT0 240
#define A2R_O_CQ1 244
#define A2R_O_CT1 248
static_assert(A2R_O_QL0 == 4 * A2_SC_QLEN0 && A2R_O_TL0 == 4 * A2_SC_TLEN0 && A2R_O_QL1 == 4 * A2_SC_QLEN1 &&
              A2R_O_TL1 == 4 * A2_SC_TLEN1 && A2R_O_QB0 == 4 * A2_SC_QB0 && A2R_O_TB0 == 4 * A2_SC_TB0 &&
              A2R_O_QB1 == 4 * A2_SC_QB1 && A2R_O_TB1 == 4 * A2_SC_TB1 && A2R_O_CQ0 == 4 * A2_SC_CQ0 &&
              A2R_O_CT0 == 4 * A2_SC_CT0 && A2R_O_CQ1 == 4 * A2_SC_CQ1 && A2R_O_CT1 == 4 * A2_SC_CT1,
              "A2R_BCAST12 spells the lanes of the packed state out");
static_assert(A2_TOP == 0x20000000u, "A2R_CONST_PAIR spells A2_TOP out");
// (values the compiler may keep on the vector unit -- in the shadow kernel the whole state comes back from memory --
// must not reach an "s" operand as they are: readfirstlane is free for the ones that are in SGPRs already)
#define A2R_UNI(x) w_uniu(x)
#define A2R_UNI64(x) (((u64)w_uniu((u32)((x) >> 32)) << 32) | w_uniu((u32)(x)))

// ---- PAIR only: a band hull reached a fence lane or an end of the wave (row J's event test): both bands are
// laid out again where they stand -- a2_replace<true> (k_align2_core.h) restated on the scalar unit.  The row
// moves by ds_bpermute (taken from the row's raw x: a hull with holes, or two hulls that met at the boundary,
// come out whole), the tape's K fields change from this iteration on, the lanes of the next row go to the
// mask register the row wrote.  Declined -- a cell at an end of a sequence or of a window, a long snake, a
// stretch that must not (row 0: the top bit of `nrep`), bands that no longer fit: out through the row's exit.
#define A2R_RELAY_PAIR                                                                                   \
    ".La2r_ev0_%=:\n\t"                                                                                  \
    "s_mov_b32 %[ret], 0\n\t"                                                                            \
    "s_branch .La2r_relay_%=\n"                                                                          \
    ".La2r_ev1_%=:\n\t"                                                                                  \
    "s_mov_b32 %[ret], 1\n\t"                                                                            \
    "s_branch .La2r_relay_%=\n"                                                                          \
    ".La2r_ev2_%=:\n\t"                                                                                  \
    "s_mov_b32 %[ret], 2\n\t"                                                                            \
    "s_branch .La2r_relay_%=\n"                                                                          \
    ".La2r_ev3_%=:\n\t"                                                                                  \
    "s_mov_b32 %[ret], 3\n"                                                                              \
    ".La2r_relay_%=:\n\t"                                                                                \
    "s_bitcmp1_b32 %[nrep], 31\n\t"                                                                      \
    "s_cbranch_scc1 .La2r_decl_%=\n\t"                                                                   \
    "s_or_b64 %[u], %[fin], %[big]\n\t"                                                                  \
    "s_cbranch_scc1 .La2r_decl_%=\n\t"                                                                   \
    "s_andn2_b64 %[c1], %[in], %[z1]\n\t"                                                                \
    "s_and_b64 %[u], %[in], %[z1]\n\t"                                                                   \
    "s_ff1_i32_b64 %[p0], %[c1]\n\t"                                                                     \
    "s_flbit_i32_b64 %[p1], %[c1]\n\t"                                                                   \
    "s_ff1_i32_b64 %[r0], %[u]\n\t"                                                                      \
    "s_flbit_i32_b64 %[r1], %[u]\n\t"                                                                    \
    "s_add_u32 %[p1], %[p1], %[p0]\n\t"                                                                  \
    "s_sub_u32 %[p1], 64, %[p1]\n\t"                                                                     \
    "s_add_u32 %[r1], %[r1], %[r0]\n\t"                                                                  \
    "s_sub_u32 %[r1], 64, %[r1]\n\t"                                                                     \
    "s_add_u32 %[r2], %[p1], %[r1]\n\t"                                                                  \
    "s_sub_i32 %[r2], 62, %[r2]\n\t"                                                                     \
    "s_cmp_lt_i32 %[r2], 0\n\t"                                                                          \
    "s_cbranch_scc1 .La2r_decl_%=\n\t"                                                                   \
    A2R_BCAST12                                                                                          \
    "s_lshr_b32 %[r3], %[r2], 2\n\t"                                                                     \
    "s_lshr_b32 %[r2], %[r2], 1\n\t"                                                                     \
    "s_add_u32 vcc_lo, %[r3], %[p1]\n\t"                                                                 \
    "s_add_u32 vcc_lo, vcc_lo, 1\n\t"                                                                    \
    "s_lshr_b32 vcc_hi, %[r2], 1\n\t"                                                                    \
    "s_add_u32 vcc_hi, vcc_hi, vcc_lo\n\t"                                                               \
    "s_add_u32 vcc_lo, vcc_lo, %[r2]\n\t"                                                                \
    "s_and_b32 %[r2], m0, 1\n\t"                                                                         \
    "s_sub_u32 %[p0], %[p0], %[r2]\n\t"                                                                  \
    "s_sub_u32 %[p0], %[p0], %[r3]\n\t"                                                                  \
    "s_sub_u32 %[r0], %[r0], %[r2]\n\t"                                                                  \
    "s_sub_u32 %[r0], %[r0], vcc_lo\n\t"                                                                 \
    "s_lshl_b64 %[z1], -1, vcc_hi\n\t"                                                                   \
    "s_sub_u32 %[ln0], vcc_hi, 1\n\t"                                                                    \
    "s_lshl_b64 %[c1], 3, %[ln0]\n\t"                                                                    \
    "s_or_b64 %[f1], %[c1], 1\n\t"                                                                       \
    "s_mov_b64 %[f0], %[c1]\n\t"                                                                         \
    "s_bitset1_b64 %[f0], 63\n\t"                                                                        \
    "s_add_u32 vcc_hi, %[r3], %[r2]\n\t"                                                                 \
    "s_bfm_b64 %[in], %[p1], vcc_hi\n\t"                                                                 \
    "s_add_u32 vcc_hi, vcc_lo, %[r2]\n\t"                                                                \
    "s_bfm_b64 %[u], %[r1], vcc_hi\n\t"                                                                  \
    "s_or_b64 %[in], %[in], %[u]\n\t"                                                                    \
    "s_add_u32 %[p1], %[p1], 1\n\t"                                                                      \
    "s_bfm_b64 %[c1], %[p1], %[r3]\n\t"                                                                  \
    "s_add_u32 %[r1], %[r1], 1\n\t"                                                                      \
    "s_bfm_b64 %[u], %[r1], vcc_lo\n\t"                                                                  \
    "s_or_b64 %[c1], %[c1], %[u]\n\t"                                                                    \
    "s_bitcmp1_b32 %[ret], 0\n\t"                                                                        \
    "s_cselect_b64 %[sa], %[c1], %[sa]\n\t"                                                              \
    "s_cselect_b64 %[sb], %[sb], %[c1]\n\t"                                                              \
    "s_lshl_b32 vcc_hi, %[p0], 1\n\t"                                                                    \
    "s_add_u32 %[kb0], %[kb0], vcc_hi\n\t"                                                               \
    "s_lshl_b32 vcc_hi, %[r0], 1\n\t"                                                                    \
    "s_add_u32 %[kb1], %[kb1], vcc_hi\n\t"                                                               \
    "s_lshl_b32 %[p0], %[p0], 2\n\t"                                                                     \
    "s_lshl_b32 %[r0], %[r0], 2\n\t"                                                                     \
    "s_add_u32 %[nrep], %[nrep], 1\n\t"                                                                  \
    "v_add_u32 %[t2], %[p0], %[l4]\n\t"                                                                  \
    "s_mov_b64 exec, %[z1]\n\t"                                                                          \
    "v_add_u32 %[t2], %[r0], %[l4]\n\t"                                                                  \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "ds_bpermute_b32 %[cinc], %[t2], %[x]\n\t"                                                           \
    "s_lshl_b64 %[u], -1, m0\n\t"                                                                        \
    "s_mov_b64 exec, %[u]\n\t"                                                                           \
    "v_mov_b32 %[k0], %[kb0]\n\t"                                                                        \
    "v_mov_b32 %[k1], %[kb1]\n\t"                                                                        \
    "s_mov_b64 exec, -1\n\t"                                                                             \
    "s_mov_b64 %[t], 0\n\t"                                                                              \
    "s_waitcnt lgkmcnt(1)\n\t"                                                                           \
    "s_branch .La2r_const_%=\n"                                                                          \
    ".La2r_decl_%=:\n\t"                                                                                 \
    "s_bitcmp1_b32 %[ret], 0\n\t"                                                                        \
    "s_cbranch_scc1 .La2r_outo_%=\n\t"                                                                   \
    "s_branch .La2r_oute_%=\n"                                                                            \
    ".La2r_lk0_%=:\n"                                                                                     \
    ".La2r_lk2_%=:\n\t"                                                                                   \
    "s_branch .La2r_oute_%=\n"                                                                            \
    ".La2r_lk1_%=:\n"                                                                                     \
    ".La2r_lk3_%=:\n\t"                                                                                   \
    "s_branch .La2r_outo_%=\n"

// ---- SINGLE: the rows asked for are done.  With the neighbour parked that is a look-up (a2_fast: every
// A2_LOOK_EVERY iterations): has the running band become narrow enough for the neighbour -- lanes from the lowest
// to the highest of `in`, plus one, <= join_at, i.e. leading + trailing zeros >= 65 - join_at = `thr`?  If not,
// and the row had no snake of >= 255 bases, the next min(`more`, A2_LOOK_EVERY) rows run without leaving the
// statement (11.8 M such look-ups per launch were a round trip of ~150 instructions each).
#define A2R_LOOK(J, OUT)                                                                                 \
    ".La2r_lk" J "_%=:\n\t"                                                                               \
    "s_cmp_eq_u32 %[more], 0\n\t"                                                                        \
    "s_cbranch_scc1 " OUT "\n\t"                                                                         \
    "s_cmp_lg_u64 %[big], 0\n\t"                                                                         \
    "s_cbranch_scc1 " OUT "\n\t"                                                                         \
    "s_ff1_i32_b64 %[p0], %[in]\n\t"                                                                     \
    "s_flbit_i32_b64 %[p1], %[in]\n\t"                                                                   \
    "s_add_u32 %[p0], %[p0], %[p1]\n\t"                                                                  \
    "s_cmp_ge_u32 %[p0], %[thr]\n\t"                                                                     \
    "s_cbranch_scc1 " OUT "\n\t"                                                                         \
    "s_min_u32 %[rem], %[more], " A2R_STR(A2_LOOK_EVERY) "\n\t"                                          \
    "s_sub_u32 %[more], %[more], %[rem]\n\t"                                                             \
    "s_branch .La2r_c" J "_%=\n"

// ---- SINGLE: a row's event leaves the stream (a2_fast centres the band: 86 times in 7 M iterations)
#define A2R_RELAY_SINGLE                                                                                 \
    ".La2r_ev0_%=:\n\t"                                                                                  \
    "s_branch .La2r_oute_%=\n"                                                                           \
    ".La2r_ev1_%=:\n\t"                                                                                  \
    "s_branch .La2r_outo_%=\n"                                                                           \
    ".La2r_ev2_%=:\n\t"                                                                                  \
    "s_branch .La2r_oute_%=\n"                                                                           \
    ".La2r_ev3_%=:\n\t"                                                                                  \
    "s_branch .La2r_outo_%=\n"                                                                            \
    A2R_LOOK("0", ".La2r_oute_%=") A2R_LOOK("1", ".La2r_outo_%=") A2R_LOOK("2", ".La2r_oute_%=") A2R_LOOK("3", ".La2r_outo_%=")

#define A2R_BODY(BEST, HULL, NRUN, CONST, RELAY)                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                           \
    "s_mov_b32 %[sm0], m0\n\t"                                                                           \
    "s_and_b32 m0, %[it], 63\n\t"                                                                        \
    "s_andn2_b32 %[itb], %[it], 63\n\t"                                                                  \
    "s_mov_b64 %[sb], %[sa]\n\t"                                                                         \
    "s_mov_b64 %[big], 0\n\t"                                                                            \
    CONST                                                                                                \
    "s_and_b32 %[p0], %[it], 3\n\t"                                                                      \
    "s_cmp_eq_u32 %[p0], 1\n\t"                                                                          \
    "s_cbranch_scc1 .La2r_r1_%=\n\t"                                                                     \
    "s_cmp_eq_u32 %[p0], 2\n\t"                                                                          \
    "s_cbranch_scc1 .La2r_r2_%=\n\t"                                                                     \
    "s_cmp_eq_u32 %[p0], 3\n\t"                                                                          \
    "s_cbranch_scc1 .La2r_r3_%=\n"                                                                       \
    A2R_EVEN("0", ".La2r_ev", "%[sel0]", A2R_NOP2, A2R_NOP2, BEST, HULL, NRUN)                           \
    A2R_ODD("1", ".La2r_ev", "%[sel1]", A2R_NOP2, A2R_NOP2, BEST, HULL, NRUN)                            \
    A2R_EVEN("2", ".La2r_ev", "%[sel2]", A2R_NOP2, A2R_NOP2, BEST, HULL, NRUN)                           \
    A2R_ODD("3", ".La2r_ev", "%[sel3]", A2R_F1_STORE, A2R_F2_STORE, BEST, HULL, NRUN)                    \
    "s_branch .La2r_r0_%=\n"                                                                             \
    A2R_EVEN_FAR("0", HULL)                                                                              \
    A2R_ODD_FAR("1", HULL)                                                                               \
    A2R_EVEN_FAR("2", HULL)                                                                              \
    A2R_ODD_FAR("3", HULL)                                                                               \
    A2R_REC                                                                                              \
    RELAY                                                                                                \
    ".La2r_oute_%=:\n\t"                                                                                 \
    "s_mov_b64 %[orow], %[sa]\n\t"                                                                       \
    "s_mov_b64 %[sa], %[sb]\n\t"                                                                         \
    "s_branch .La2r_out_%=\n"                                                                            \
    ".La2r_outo_%=:\n\t"                                                                                 \
    "s_mov_b64 %[orow], %[sb]\n"                                                                         \
    ".La2r_out_%=:\n\t"                                                                                  \
    "s_add_u32 %[it], %[itb], m0\n\t"                                                                    \
    "s_mov_b32 m0, %[sm0]\n\t"

// the sums of the lanes' cell counters into h.cells0 / h.cells1; the counters start over (the end of a2_fast).
// PAIR: a lane's counter holds the cells it computed for track 0 in its low half and those for track 1 in its
// high half (A2R_CONST_PAIR's `cinc`), so lanes change hands without a fold; a half cannot overflow -- a call
// of a2_fast runs fewer iterations than the tape ring holds (<= 32768).
template <bool PAIR>
W_FN void a2_fold_cells(A2Hot &h, vu &vcnt) {
    if (PAIR) {
        h.cells0 += w_readlaneu(w_prefix_add(vcnt & 0xffffu), 63);
        h.cells1 += w_readlaneu(w_prefix_add(vcnt >> 16), 63);
    } else {
        h.cells0 += w_readlaneu(w_prefix_add(vcnt), 63);
    }
    vcnt = 0u;
}

// ---------------------------------------------------------------------------------------
// The LDS windows.  A row reads 16 bases of the query and 16 of the target per cell, at
// positions that creep forward by a base or two per row -- 2 x 64-lane gathers per iteration,
// and with 32 wavefronts per CU the vector-memory path was what a wavefront waited for (52 % of
// its cycles parked at the rows' s_waitcnt, profiles/r05_*).  So the stretch of either sequence
// a track is walking sits in the LDS: per track a window of 4096 query and 4096 target bases
// (256 words each, A2W_*), filled with one 16-byte read per lane, read by the rows with
// ds_read2_b32.  A window only ever moves forward (the smallest x and y of a band never
// decrease).  The rows leave when a cell comes within A2W_MARGIN bases of a window's end -- the
// same compare that sees a cell reach the end of its sequence, against min(L, the window's
// limit) -- and the window is filled again from where the band stands.  What may read anywhere
// -- a snake beyond its first 16 bases, the wide rows -- keeps reading global memory.
// Word bases of the four windows live in LDS words A2W_HDR.. between two calls of a2_fast (the
// event loop marks a track's windows invalid when it fetches a new alignment or lets the wide
// rows use the LDS: a2_win_invalidate).
// ---------------------------------------------------------------------------------------
#define A2W_WORDS 256
#define A2W_MARGIN 128
#define A2W_USABLE 252   // words of a window the rows may count on (the last window ends at A2W_HDR)
static_assert(A2W_HDR == 3 * A2W_WORDS + A2W_USABLE && A2_LDS_WORDS >= A2W_HDR + 4, "windows, the last one four words short for their header");

// State of the stream that outlives a stretch
struct A2RowsV {
    vi tdn, tup;   // the previous row shifted one lane up / down; lane 0 / lane 63 hold A2_NEG for good
    vu vcnt;       // cells per lane since the last a2_fold_cells
    u32 wq0, wt0, wq1, wt1;  // first word (index into words[]) of the windows of track 0 / 1, or A2W_INVALID
};

W_FN void a2_win_load(A2RowsV &rv) {
    const u32 *l = w_lds();
    rv.wq0 = w_uniu(w_lds_bcast(l, A2W_HDR + 0)); rv.wt0 = w_uniu(w_lds_bcast(l, A2W_HDR + 1));
    rv.wq1 = w_uniu(w_lds_bcast(l, A2W_HDR + 2)); rv.wt1 = w_uniu(w_lds_bcast(l, A2W_HDR + 3));
}
W_FN void a2_win_store(const A2RowsV &rv) {
    u32 *l = w_lds();
    const vi lane = w_lane();
    const vu v = w_selu(1ull, w_selu(2ull, w_selu(4ull, (vu)rv.wt1, (vu)rv.wq1), (vu)rv.wt0), (vu)rv.wq0);
    W_WHERE(0xfull) { w_lds_store(l, (vu)(A2W_HDR + lane), v); }
}

// Fill the two windows of track TI (its lanes: `zone`) from where its band stands: the smallest
// x and y of its last row (a track without rows: 0).  `src`: a lane of the zone (its per-lane
// constants are the track's).
template <int TI>
W_FN void a2_win_fill(const u32 *words, const A2HotV &hv, u64 zone, int src, u32 &wq, u32 &wt) {
    const u64 hull = w_ballot(hv.vx >= 0) & zone;
    const int xmin = hull ? max(0, w_reduce_min(w_sel(hull, 0x7fffffff, hv.vx))) : 0;
    // (an odd row has y = x + vnegk - 1, an even one y = x + vnegk: the smaller)
    const int ymin = hull ? max(0, w_reduce_min(w_sel(hull, 0x7fffffff, hv.vx + hv.vnegk - 1))) : 0;
    wq = (w_readlaneu(hv.vqb, src) + (u32)xmin) >> 4;
    wt = (w_readlaneu(hv.vtb, src) + (u32)ymin) >> 4;
    u32 *l = w_lds();
    const vu lane4 = (vu)w_lane() << 2;
    vu a0 = w_load32(words, wq + lane4), a1 = w_load32(words, wq + lane4 + 1u);
    vu a2 = w_load32(words, wq + lane4 + 2u), a3 = w_load32(words, wq + lane4 + 3u);
    vu b0 = w_load32(words, wt + lane4), b1 = w_load32(words, wt + lane4 + 1u);
    vu b2 = w_load32(words, wt + lane4 + 2u), b3 = w_load32(words, wt + lane4 + 3u);
    const vu at = (vu)(TI * 2 * A2W_WORDS) + lane4;
    w_lds_store(l, at, a0); w_lds_store(l, at + 1u, a1); w_lds_store(l, at + 2u, a2); w_lds_store(l, at + 3u, a3);
    // (the target window of track 1 is the LDS's last: its last four words are the header's)
    W_WHERE(TI == 1 ? ~(1ull << 63) : ~0ull) {
        w_lds_store(l, at + A2W_WORDS, b0); w_lds_store(l, at + (A2W_WORDS + 1u), b1);
        w_lds_store(l, at + (A2W_WORDS + 2u), b2); w_lds_store(l, at + (A2W_WORDS + 3u), b3);
    }
}

// One stretch of rows through the stream: from h.it until a row raises an event the stream does not
// resolve itself, or `it_end`.  `xrow`: x of every lane after the last row (what hv.vx holds of it inside the
// hull).  PAIR: the stream lays the bands out again by itself (h.split / zone1 / forbid_* / kb0 / kb1 / in / act
// come back as the last layout has them, h.n_replace counts; `relay` = false forbids it: a stretch that begins
// with a track's row 0) and makes its per-lane constants itself, of the scalars in `sc` (a2_fast's packed state);
// what the code around the rows keeps per lane of a layout (hv.vnegk .. hv.vtop) is made again here, so none of
// it stays alive across the stream -- with it the function needed more than its 64 vector registers, and every
// call of a2_fast paid for the spills.
template <bool PAIR>
W_FN void a2_rows_stream(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, A2RowsV &rv, const u32 *words, u32 *cells,
                         u32 *recs, u32 ring, int band, u32 it_end, vi &xrow, vu &sc, bool relay, u32 more,
                         u32 thr) {
    const vi lane = w_lane();
    // In global memory the query base of a cell is x + qb, the target base x + tb + vnegk (- 1 in an odd row:
    // even rows have y = x + vnegk, odd ones y = x + vnegk - 1); in the LDS (as a base index: 16 per word,
    // window w of track t from word (2 t + w) 256 on) x + cq and x + ct (- 1); xq / xt: the x from which on a
    // window's end is nearer than the margin.
    const bool t0 = PAIR || h.kb0 != A2_INVALID;   // (a track running alone plays track 0, whichever it is)
    const u32 aq0 = 16u * (0u * A2W_WORDS - rv.wq0), at0 = 16u * (1u * A2W_WORDS - rv.wt0);
    const u32 aq1 = 16u * (2u * A2W_WORDS - rv.wq1), at1 = 16u * (3u * A2W_WORDS - rv.wt1);
    const int room = 16 * A2W_USABLE - A2W_MARGIN;
    static_assert(A2R_ROOMQ0 == 16 * A2W_USABLE - A2W_MARGIN && A2R_ROOMT0 == A2R_ROOMQ0 + 16 * A2W_WORDS &&
                  A2R_ROOMQ1 == A2R_ROOMQ0 + 32 * A2W_WORDS && A2R_ROOMT1 == A2R_ROOMQ0 + 48 * A2W_WORDS,
                  "A2R_CONST_PAIR spells the windows' limits out");
    const vu l4 = (vu)lane << 2;
    const vi neg = A2_NEG;
    u64 sa = h.act, sb, in, fin, big, t, c1, u, orow;
    u32 b0 = (u32)h.best0, b1 = (u32)h.best1;
    u32 rem = it_end - h.it - 1u, it = h.it, itb, p0, p1, sm0;
    u32 coff = ((it >> 2) & ((ring >> 2) - 1u)) << 8;            // byte offset of the current group of 4's cell words
    u32 roff = ((it & ~63u) & (ring - 1u)) << 4;                 // ... of the current block of 64's records
    const u32 cmaskb = ((ring >> 2) << 8) - 1u, rmaskb = (ring << 4) - 1u;
    vu vm, x, qa, ta, t1, t2, key, pm;
    vi le, lo;
    if (PAIR) {
        // where the windows stand, into the packed state beside the tracks' other scalars (A2R_CONST_PAIR)
        w_pack_put<A2_SC_CQ0>(sc, w_pack_get<A2_SC_QB0>(sc) + aq0);
        w_pack_put<A2_SC_CT0>(sc, w_pack_get<A2_SC_TB0>(sc) + at0);
        w_pack_put<A2_SC_CQ1>(sc, w_pack_get<A2_SC_QB1>(sc) + aq1);
        w_pack_put<A2_SC_CT1>(sc, w_pack_get<A2_SC_TB1>(sc) + at1);
        u64 z1 = A2R_UNI64(h.zone1), f0 = A2R_UNI64(h.forbid_to0), f1 = A2R_UNI64(h.forbid_to1);
        // (`nrep`: the re-layouts so far; its top bit set: none allowed in this stretch)
        u32 kb0 = A2R_UNI(h.kb0), kb1 = A2R_UNI(h.kb1), ret = 4u, r0, r1, r2, r3;
        u32 nrep = A2R_UNI(h.n_replace | (relay ? 0u : 0x80000000u));
        u32 ln0 = w_uniu((u32)(h.split - 1) & 63u);  // (the compiler keeps `split` on the vector unit)
        vu cq, ct, cqg, ctg, lwe, lwo, cke, cko, cinc;
        asm volatile(A2R_BODY(A2R_BEST_PAIR, A2R_HULL_PAIR, "2", A2R_CONST_PAIR, A2R_RELAY_PAIR)
                     : [vx] "+v"(hv.vx), [acc] "+v"(hv.vacc), [mlo] "+v"(hv.rc_mlo), [mhi] "+v"(hv.rc_mhi),
                       [k0] "+v"(rc_k0), [k1] "+v"(rc_k1), [cnt] "+v"(rv.vcnt), [tdn] "+v"(rv.tdn), [tup] "+v"(rv.tup),
                       [sc] "+v"(sc),
                       [vm] "=&v"(vm), [x] "=&v"(x), [qa] "=&v"(qa), [ta] "=&v"(ta), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [key] "=&v"(key), [pm] "=&v"(pm),
                       [cq] "=&v"(cq), [ct] "=&v"(ct), [cqg] "=&v"(cqg), [ctg] "=&v"(ctg),
                       [le] "=&v"(le), [lo] "=&v"(lo), [lwe] "=&v"(lwe), [lwo] "=&v"(lwo), [cke] "=&v"(cke),
                       [cko] "=&v"(cko), [cinc] "=&v"(cinc),
                       [sa] "+s"(sa), [b0] "+s"(b0), [b1] "+s"(b1), [rem] "+s"(rem), [it] "+s"(it),
                       [coff] "+s"(coff), [roff] "+s"(roff),
                       [z1] "+s"(z1), [f0] "+s"(f0), [f1] "+s"(f1), [kb0] "+s"(kb0), [kb1] "+s"(kb1),
                       [ln0] "+s"(ln0), [nrep] "+s"(nrep), [ret] "+s"(ret),
                       [sb] "=&s"(sb), [in] "=&s"(in), [fin] "=&s"(fin), [big] "=&s"(big), [t] "=&s"(t),
                       [c1] "=&s"(c1), [u] "=&s"(u), [p0] "=&s"(p0), [p1] "=&s"(p1), [itb] "=&s"(itb), [orow] "=&s"(orow),
                       [sm0] "=&s"(sm0), [r0] "=&s"(r0), [r1] "=&s"(r1), [r2] "=&s"(r2), [r3] "=&s"(r3)
                     : [l4] "v"(l4), [neg] "v"(neg), [band] "s"(band),
                       [words] "s"(words), [cells] "s"(cells), [recs] "s"(recs),
                       [cmaskb] "s"(cmaskb), [rmaskb] "s"(rmaskb),
                       [sel0] "s"(0x03020104u), [sel1] "s"(0x03020400u), [sel2] "s"(0x03040100u),
                       [sel3] "s"(0x04020100u)
                     : "vcc", "scc", "memory", "m0", "v52", "v53", "v54", "v55");
        // the layout as the stream left it, and what the code around the rows keeps per lane of it (a2_replace's
        // last lines), from the scalars
        h.n_replace = nrep & 0x7fffffffu;
        h.zone1 = z1; h.forbid_to0 = f0; h.forbid_to1 = f1;
        h.split = (int)ln0 + 1;
        h.kb0 = kb0; h.kb1 = kb1;
        hv.vnegk = w_sel(h.zone1, 1 - (int)h.kb0, 1 - (int)h.kb1) - 2 * lane;
        hv.vqlen = w_sel(h.zone1, (int)w_pack_get<A2_SC_QLEN0>(sc), (int)w_pack_get<A2_SC_QLEN1>(sc));
        hv.vtlen = w_sel(h.zone1, (int)w_pack_get<A2_SC_TLEN0>(sc), (int)w_pack_get<A2_SC_TLEN1>(sc));
        hv.vqb = w_selu(h.zone1, w_pack_get<A2_SC_QB0>(sc), w_pack_get<A2_SC_QB1>(sc));
        hv.vtb = w_selu(h.zone1, w_pack_get<A2_SC_TB0>(sc), w_pack_get<A2_SC_TB1>(sc));
        hv.vtop = w_selu(h.zone1, 0u, A2_TOP);
    } else {
        const vu cqg = hv.vqb;
        const vu ctg = hv.vtb + (vu)hv.vnegk;
        const vu cq = cqg + (t0 ? aq0 : aq1);
        const vu ct = ctg + (t0 ? at0 : at1);
        const vi xq = (t0 ? room : room + 32 * A2W_WORDS) - (vi)cq;
        const vi xt = (t0 ? room + 16 * A2W_WORDS : room + 48 * A2W_WORDS) - (vi)ct;
        const vi tn = hv.vtlen - hv.vnegk;
        le = w_min(hv.vqlen, tn); lo = w_min(hv.vqlen, tn + 1);
        const vi xw = w_min(xq, xt);
        const vi lwe = w_min(le, xw), lwo = w_min(lo, xw);
        const vu cke = (vu)hv.vnegk + hv.vtop, cko = cke - 1u;
        const vu cinc = 1u;
        asm volatile(A2R_BODY(A2R_BEST_SINGLE, A2R_HULL_SINGLE, "1", "", A2R_RELAY_SINGLE)
                     : [vx] "+v"(hv.vx), [acc] "+v"(hv.vacc), [mlo] "+v"(hv.rc_mlo), [mhi] "+v"(hv.rc_mhi),
                       [k0] "+v"(rc_k0), [k1] "+v"(rc_k1), [cnt] "+v"(rv.vcnt), [tdn] "+v"(rv.tdn), [tup] "+v"(rv.tup),
                       [vm] "=&v"(vm), [x] "=&v"(x), [qa] "=&v"(qa), [ta] "=&v"(ta), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [key] "=&v"(key), [pm] "=&v"(pm),
                       [sa] "+s"(sa), [b0] "+s"(b0), [rem] "+s"(rem), [it] "+s"(it),
                       [coff] "+s"(coff), [roff] "+s"(roff), [more] "+s"(more),
                       [sb] "=&s"(sb), [in] "=&s"(in), [fin] "=&s"(fin), [big] "=&s"(big), [t] "=&s"(t),
                       [c1] "=&s"(c1), [p0] "=&s"(p0), [p1] "=&s"(p1), [itb] "=&s"(itb), [orow] "=&s"(orow),
                       [sm0] "=&s"(sm0)
                     : [cq] "v"(cq), [ct] "v"(ct), [cqg] "v"(cqg), [ctg] "v"(ctg), [le] "v"(le), [lo] "v"(lo),
                       [lwe] "v"(lwe), [lwo] "v"(lwo), [cke] "v"(cke), [cko] "v"(cko), [cinc] "v"(cinc), [l4] "v"(l4),
                       [neg] "v"(neg), [band] "s"(band), [thr] "s"(thr),
                       [f0] "s"(h.forbid_to0), [f1] "s"(h.forbid_to1),
                       [words] "s"(words), [cells] "s"(cells), [recs] "s"(recs),
                       [cmaskb] "s"(cmaskb), [rmaskb] "s"(rmaskb), [kb0] "s"(h.kb0), [kb1] "s"(h.kb1),
                       [sel0] "s"(0x03020104u), [sel1] "s"(0x03020400u), [sel2] "s"(0x03040100u),
                       [sel3] "s"(0x04020100u)
                     : "vcc", "scc", "memory", "m0", "v52", "v53", "v54", "v55");
        (void)u; (void)b1; (void)sc; (void)relay;
    }
    if (PAIR) {
        (void)more; (void)thr;
    }
    // What the stream hands back: `sa` the lanes of the next row, `orow` those of the last one
    // (the exits put them there, whichever way the two mask registers stood); `t`: the band hull
    // on a forbidden lane, or a cell at x >= min(L, a window's limit).
    h.it = it;
    h.act = sa;
    h.act_row = orow;
    h.in = in; h.big = big;
    h.best0 = (int)b0;
    if (PAIR) h.best1 = (int)b1;
    if (fin == 0ull) {
        h.fin = 0ull;
        h.ev = t | big;
    } else {
        // The cells at x >= L reached an end of a sequence; a track with any of the others has
        // its windows filled again before its next row.
        const bool last_even = ((it - 1u) & 1u) == 0u;
        const u64 ended = w_ballot((vi)x >= (last_even ? le : lo)) & fin;
        const u64 moved = fin & ~ended;
        if (moved) {
            if (PAIR) {
                if (moved & ~h.zone1) rv.wq0 = A2W_INVALID;
                if (moved & h.zone1) rv.wq1 = A2W_INVALID;
            } else if (t0) {
                rv.wq0 = A2W_INVALID;
            } else {
                rv.wq1 = A2W_INVALID;
            }
        }
        h.fin = ended;
        h.ev = ended | big | (in & (last_even ? h.forbid_to1 : h.forbid_to0));
    }
    hv.vm = vm;
    xrow = (vi)x;
}

// The rows of a stretch.  `head`: the first stretch after the wavefront's event loop placed the
// tracks -- the only one that can begin with a track's row 0.  That row is the one case in which
// a lane OUTSIDE the bands computes a plausible x: before it the track's lanes hold a single 0
// (V[1] of the reference's zeroed array, DW_banded.c:153) that is no hull, and the lane above
// it reads it as its V[k-1].  The stream does not mask the filter with the row's lanes, so a
// row 0 runs alone and what it hands back is cut to the row's lanes here.
template <bool PAIR>
W_FN void a2_rows_asm(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, A2RowsV &rv, const u32 *words, u32 *cells,
                      u32 *recs, u32 ring, int band, u32 it_end, bool head, vu &sc, u32 it_last, int join_at) {
    bool row0 = false;
    if (head) {
        row0 = PAIR ? (w_popc(h.act & ~h.zone1) == 1 || w_popc(h.act & h.zone1) == 1) : w_popc(h.act) == 1;
    }
#if defined(A2_DBG_COUNT)
    // (experiments only: what the `parkings` statistic counts instead -- 1: window fills, 2: calls of the stream)
#define A2_DBG_TICK(n) w_pack_put<A2_SC_NPARK>(sc, w_pack_get<A2_SC_NPARK>(sc) + (n))
    if (A2_DBG_COUNT == 2) A2_DBG_TICK(1u);
    if (A2_DBG_COUNT == 1) {
        if (PAIR) A2_DBG_TICK((rv.wq0 == A2W_INVALID ? 1u : 0u) + (rv.wq1 == A2W_INVALID ? 1u : 0u));
        else A2_DBG_TICK(((h.kb0 != A2_INVALID ? rv.wq0 : rv.wq1) == A2W_INVALID) ? 1u : 0u);
    }
#endif
    // the windows of the tracks that run
    if (PAIR) {
        if (rv.wq0 == A2W_INVALID) a2_win_fill<0>(words, hv, ~h.zone1, 0, rv.wq0, rv.wt0);
        if (rv.wq1 == A2W_INVALID) a2_win_fill<1>(words, hv, h.zone1, 63, rv.wq1, rv.wt1);
    } else if (h.kb0 != A2_INVALID) {
        if (rv.wq0 == A2W_INVALID) a2_win_fill<0>(words, hv, ~0ull, 0, rv.wq0, rv.wt0);
    } else {
        if (rv.wq1 == A2W_INVALID) a2_win_fill<1>(words, hv, ~0ull, 0, rv.wq1, rv.wt1);
    }
    vi x;
    // (SINGLE, the neighbour parked: the look-ups behind `it_end` are the stream's own -- A2R_LOOK)
    const bool looks = !PAIR && !row0 && join_at > 0 && (int)(it_last - it_end) > 0;
#if defined(A2_DBG_NORELAY)
    a2_rows_stream<PAIR>(h, hv, rc_k0, rc_k1, rv, words, cells, recs, ring, band, row0 ? h.it + 1u : it_end, x, sc, false,
                         0u, 0u);
#else
    a2_rows_stream<PAIR>(h, hv, rc_k0, rc_k1, rv, words, cells, recs, ring, band, row0 ? h.it + 1u : it_end, x, sc,
                         !row0, looks ? it_last - it_end : 0u, (u32)(65 - join_at));
#endif
    // The other case the stream gets wrong: two bands laid out with no lane to spare are neighbours,
    // and when both reach the boundary between the tracks their filter masks merge into one run --
    // two runs in all, which the stream takes for one hull per track, holes and all.  Both lanes
    // around the boundary are forbidden to a hull, so such a row always ends its stretch, and
    // its hulls are made again here.
    const u64 fence = PAIR ? (3ull << ((h.split - 1) & 63)) : 0ull;
    if (row0 || (PAIR && (h.in & fence) == fence)) {  // (after a row 0 the caller's loop goes on from here)
        const bool even = ((h.it - 1u) & 1u) == 0u;
        const u64 in = h.in & h.act_row;
        u64 hull;
        if (PAIR) {
            const u64 in0 = in & ~h.zone1, in1 = in & h.zone1;
            hull = w_lanes(w_lowest(in0), w_span(in0)) | w_lanes(w_lowest(in1), w_span(in1));
        } else {
            hull = w_lanes(w_lowest(in), w_span(in));
        }
        hv.vx = w_sel(hull, A2_NEG, x);
        h.act = even ? (hull | (hull >> 1)) : (hull | (hull << 1));
        h.in = in;
        h.fin &= h.act_row;
        h.ev = h.fin | h.big | (in & (even ? h.forbid_to1 : h.forbid_to0));
    }
}

// the beginning and the end of a call of a2_fast: the windows' word bases from and back to the LDS
W_FN void a2_rows_begin(A2RowsV &rv) {
    rv.tdn = A2_NEG; rv.tup = A2_NEG; rv.vcnt = 0u;
    a2_win_load(rv);
}
W_FN void a2_rows_done(A2RowsV &rv) { a2_win_store(rv); }
