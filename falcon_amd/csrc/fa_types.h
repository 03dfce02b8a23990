// fa_types.h -- plain data structures of a batch in HBM (no HIP in here: the kernels, the
// host engine and the lane emulator of tests/emu all read them).  See fa_internal.h for the
// layout they describe.
#pragma once
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;

#define FA_K 8                 // k-mer size (consensus.py:270 hard-wires 8)
#define FA_NKMER 65536         // 4^8
#define FA_IDX_STRIDE 65540    // u32 per pile for the CSR table (65537 used, padded)
#define FA_BAND 150            // falcon.c:624 INDEL_ALLOWENCE_2
#define FA_ALIGN_MAXCH 3       // 64-lane chunks per band row (<= 191 diagonals)
#define FA_CNS_MAX_ALN 65534   // accepted alignments per pile the MSA kernels handle (16-bit link counts, like
                               // the reference's, falcon.c:86); a deeper pile is reported, alone

struct FaSeq {
    u32 woff;   // offset into words[]
    int len;    // bases
    int pile;   // pile id
    int idx;    // index in pile, 0 = seed
};

struct FaPile {
    int first;      // global index of the seed
    int n_seq;
    int seed_len;
    int pad0;
    u64 kidx_off;   // u32 offset of this pile's CSR table
    u64 kpos_off;   // u32 offset of this pile's position list
    u64 node_off;   // node offset in nodes[] (set after the alignment stage)
    u64 node_cap;   // nodes available (multiple of 5)
    u64 out_off;    // char/int offset into out_seq / out_eqv (2*T+2 slots)
};

struct FaRange {
    int s1, e1, s2, e2;
    int ok;         // 1 = passed the falcon.c:613-619 sanity filter
    int n_hit;      // diagnostic: k-mer hits
    long long score;
};

struct FaAln {
    int dist;
    int q_e, t_e;
    int size;       // alignment columns (aln_str_size)
    int accept;     // falcon.c:629 verdict
    int n_ins;      // query-only edit rows (bounds the number of MSA levels)
    int aligned;    // 1 = the O(ND) search reached a sequence end
    int err;        // 1 = resource overflow (must not happen; checked on host)
    long long cells;// (d,k) cells evaluated
};

