// valu_cost.hip -- what each instruction of k_align2's row stream costs the vector pipe of a gfx950 SIMD.
//
// issue_rates.hip (round 2) found v_add_u32 / v_and_b32 at twice the rate of v_max_i32 / shifts / compares:
// the kernel is bound by the vector pipe's cycles, not by the instruction count, so the table the rows are
// tuned against is "SIMD cycles per wave64 instruction", per opcode.  Every kernel: 8 wavefronts per SIMD,
// ITER x 256 instructions of one kind on 8 independent registers; reports wave-instructions per CU per ns
// and the cycles per instruction per SIMD relative to the first line (v_add_u32).  The mixes at the end ask
// what a scalar instruction, an s_nop, an LDS read and an LDS atomic cost BESIDE vector work.
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/valu_cost.bin scripts/ubench/valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 1000
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(s) s s s s s s s s
#define REP32(s) REP8(s) REP8(s) REP8(s) REP8(s)

// body: 8 instructions on %0..%7 (VGPRs, read-write), %8 a VGPR input, %9 an SGPR pair (a lane mask), %10 an SGPR
#define KERNEL(name, body)                                                                          \
    __global__ __launch_bounds__(64, 8) void name(unsigned *out, int n_lds) {                        \
        extern __shared__ unsigned lds[];                                                             \
        unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5,              \
                 g = a + 6, h = a + 7, k = (blockIdx.x & 15) | 1;                                     \
        unsigned long long m = 0x00ffff0000ffff00ull ^ blockIdx.x;                                    \
        unsigned s = blockIdx.x & 31;                                                                 \
        for (int i = threadIdx.x; i < n_lds; i += 64) lds[i] = i;                                     \
        for (int i = 0; i < ITER; i++)                                                                \
            asm volatile(REP32(body)                                                                  \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)     \
                         : "v"(k), "s"(m), "s"(s) : "vcc", "scc", "memory", "s90", "s91", "s92", "s93", "m0", "v60", "v61"); \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;                           \
    }
#define V8(op) op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
               op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8\n\t"
// one-source forms: op dst, src
#define U8(op) op " %0, %1\n\t" op " %1, %2\n\t" op " %2, %3\n\t" op " %3, %4\n\t" \
               op " %4, %5\n\t" op " %5, %6\n\t" op " %6, %7\n\t" op " %7, %0\n\t"
// three-source forms: op dst, dst, %8, x
#define T8(op, x) op " %0, %0, %8, " x "\n\t" op " %1, %1, %8, " x "\n\t" op " %2, %2, %8, " x "\n\t" op " %3, %3, %8, " x "\n\t" \
                  op " %4, %4, %8, " x "\n\t" op " %5, %5, %8, " x "\n\t" op " %6, %6, %8, " x "\n\t" op " %7, %7, %8, " x "\n\t"
#define SFX8(op, sfx) op " %0, %1, %1 " sfx "\n\t" op " %1, %2, %2 " sfx "\n\t" op " %2, %3, %3 " sfx "\n\t" op " %3, %4, %4 " sfx "\n\t" \
                      op " %4, %5, %5 " sfx "\n\t" op " %5, %6, %6 " sfx "\n\t" op " %6, %7, %7 " sfx "\n\t" op " %7, %0, %0 " sfx "\n\t"
#define MOVDPP8(sfx) "v_mov_b32_dpp %0, %1 " sfx "\n\t" "v_mov_b32_dpp %1, %2 " sfx "\n\t" "v_mov_b32_dpp %2, %3 " sfx "\n\t" "v_mov_b32_dpp %3, %4 " sfx "\n\t" \
                     "v_mov_b32_dpp %4, %5 " sfx "\n\t" "v_mov_b32_dpp %5, %6 " sfx "\n\t" "v_mov_b32_dpp %6, %7 " sfx "\n\t" "v_mov_b32_dpp %7, %0 " sfx "\n\t"
#define CMP8(op, dst) op " " dst ", %0, %8\n\t" op " " dst ", %1, %8\n\t" op " " dst ", %2, %8\n\t" op " " dst ", %3, %8\n\t" \
                      op " " dst ", %4, %8\n\t" op " " dst ", %5, %8\n\t" op " " dst ", %6, %8\n\t" op " " dst ", %7, %8\n\t"

KERNEL(k_add, V8("v_add_u32"))
KERNEL(k_sub, V8("v_sub_u32"))
KERNEL(k_and, V8("v_and_b32"))
KERNEL(k_or, V8("v_or_b32"))
KERNEL(k_xor, V8("v_xor_b32"))
KERNEL(k_mov, U8("v_mov_b32"))
KERNEL(k_lshl, V8("v_lshlrev_b32"))
KERNEL(k_lshr, V8("v_lshrrev_b32"))
KERNEL(k_ashr, V8("v_ashrrev_i32"))
KERNEL(k_max_i, V8("v_max_i32"))
KERNEL(k_min_u, V8("v_min_u32"))
KERNEL(k_min3, T8("v_min3_u32", "16"))
KERNEL(k_max3, T8("v_max3_i32", "16"))
KERNEL(k_add3, T8("v_add3_u32", "-1"))
KERNEL(k_lshl_add, T8("v_lshl_add_u32", "1"))
KERNEL(k_add_lshl, T8("v_add_lshl_u32", "1"))
KERNEL(k_and_or, T8("v_and_or_b32", "3"))
KERNEL(k_lshl_or, T8("v_lshl_or_b32", "3"))
KERNEL(k_bfe, T8("v_bfe_u32", "5"))
KERNEL(k_alignbit, T8("v_alignbit_b32", "%8"))
KERNEL(k_perm, T8("v_perm_b32", "%10"))
KERNEL(k_mad24, T8("v_mad_u32_u24", "%8"))
KERNEL(k_sad, T8("v_sad_u32", "%8"))
KERNEL(k_ffbl, U8("v_ffbl_b32"))
KERNEL(k_add_lit, "v_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3\n\t"
                  "v_add_u32 %4, 0x12345, %4\n\tv_add_u32 %5, 0x12345, %5\n\tv_add_u32 %6, 0x12345, %6\n\tv_add_u32 %7, 0x12345, %7\n\t")
KERNEL(k_add_sgpr, "v_add_u32 %0, %10, %0\n\tv_add_u32 %1, %10, %1\n\tv_add_u32 %2, %10, %2\n\tv_add_u32 %3, %10, %3\n\t"
                   "v_add_u32 %4, %10, %4\n\tv_add_u32 %5, %10, %5\n\tv_add_u32 %6, %10, %6\n\tv_add_u32 %7, %10, %7\n\t")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %8\n\tv_add_u32_e64 %1, %1, %8\n\tv_add_u32_e64 %2, %2, %8\n\tv_add_u32_e64 %3, %3, %8\n\t"
                  "v_add_u32_e64 %4, %4, %8\n\tv_add_u32_e64 %5, %5, %8\n\tv_add_u32_e64 %6, %6, %8\n\tv_add_u32_e64 %7, %7, %8\n\t")
KERNEL(k_max_e64, "v_max_i32_e64 %0, %0, %8\n\tv_max_i32_e64 %1, %1, %8\n\tv_max_i32_e64 %2, %2, %8\n\tv_max_i32_e64 %3, %3, %8\n\t"
                  "v_max_i32_e64 %4, %4, %8\n\tv_max_i32_e64 %5, %5, %8\n\tv_max_i32_e64 %6, %6, %8\n\tv_max_i32_e64 %7, %7, %8\n\t")
KERNEL(k_cmp_vcc, CMP8("v_cmp_lt_i32", "vcc"))
KERNEL(k_cmp_sgpr, CMP8("v_cmp_lt_i32", "s[90:91]"))
KERNEL(k_cmp_eq_const, "v_cmp_eq_u32 vcc, 16, %0\n\tv_cmp_eq_u32 vcc, 16, %1\n\tv_cmp_eq_u32 vcc, 16, %2\n\tv_cmp_eq_u32 vcc, 16, %3\n\t"
                       "v_cmp_eq_u32 vcc, 16, %4\n\tv_cmp_eq_u32 vcc, 16, %5\n\tv_cmp_eq_u32 vcc, 16, %6\n\tv_cmp_eq_u32 vcc, 16, %7\n\t")
KERNEL(k_cndmask_sgpr, T8("v_cndmask_b32", "%9"))
KERNEL(k_mov_dpp_shr, MOVDPP8("wave_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp_ror, MOVDPP8("wave_ror:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp_row, MOVDPP8("row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_max_dpp_row, SFX8("v_max_i32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_max_dpp_bc15, SFX8("v_max_i32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf"))
KERNEL(k_add_dpp_row, SFX8("v_add_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_readlane, "v_readlane_b32 s90, %0, 5\n\tv_readlane_b32 s91, %1, 5\n\tv_readlane_b32 s92, %2, 5\n\tv_readlane_b32 s93, %3, 5\n\t"
                   "v_readlane_b32 s90, %4, %10\n\tv_readlane_b32 s91, %5, %10\n\tv_readlane_b32 s92, %6, %10\n\tv_readlane_b32 s93, %7, %10\n\t")
KERNEL(k_writelane, "v_writelane_b32 %0, %10, 5\n\tv_writelane_b32 %1, %10, 6\n\tv_writelane_b32 %2, %10, 7\n\tv_writelane_b32 %3, %10, 8\n\t"
                    "v_writelane_b32 %4, %10, 5\n\tv_writelane_b32 %5, %10, 6\n\tv_writelane_b32 %6, %10, 7\n\tv_writelane_b32 %7, %10, 8\n\t")
KERNEL(k_writelane_m0, "s_mov_b32 m0, 3\n\tv_writelane_b32 %0, %10, m0\n\tv_writelane_b32 %1, %10, m0\n\tv_writelane_b32 %2, %10, m0\n\tv_writelane_b32 %3, %10, m0\n\t"
                       "v_writelane_b32 %4, %10, m0\n\tv_writelane_b32 %5, %10, m0\n\tv_writelane_b32 %6, %10, m0\n\tv_writelane_b32 %7, %10, m0\n\t")
KERNEL(k_mbcnt, V8("v_mbcnt_lo_u32_b32"))
// ---- mixes: 8 vector instructions and something else beside each of them
#define MIX8(vop, other) vop " %0, %0, %8\n\t" other vop " %1, %1, %8\n\t" other vop " %2, %2, %8\n\t" other vop " %3, %3, %8\n\t" other \
                         vop " %4, %4, %8\n\t" other vop " %5, %5, %8\n\t" other vop " %6, %6, %8\n\t" other vop " %7, %7, %8\n\t" other
KERNEL(k_mix_max_sadd, MIX8("v_max_i32", "s_add_u32 s90, s90, %10\n\t"))
KERNEL(k_mix_max_2sadd, MIX8("v_max_i32", "s_add_u32 s90, s90, %10\n\ts_and_b64 s[92:93], s[92:93], %9\n\t"))
KERNEL(k_mix_add_2sadd, MIX8("v_add_u32", "s_add_u32 s90, s90, %10\n\ts_and_b64 s[92:93], s[92:93], %9\n\t"))
KERNEL(k_mix_max_nop, MIX8("v_max_i32", "s_nop 1\n\t"))
KERNEL(k_mix_max_add, "v_max_i32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_max_i32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                      "v_max_i32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_max_i32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t")
// LDS beside vector work: a read per 8 vector instructions (the rows: 2 per ~39), addresses in the wave's own 1 KB
KERNEL(k_mix_max_dsread, V8("v_max_i32") "v_and_b32 %7, 0x3fc, %7\n\tds_read2_b32 v[60:61], %7 offset1:1\n\t")
// an LDS atomic maximum by 64 / 6 / 2 lanes on ONE word (same-address conflicts), and the read behind it
KERNEL(k_mix_max_dsmax64, V8("v_max_i32") "v_mov_b32 v60, 0\n\tds_max_i32 v60, %0\n\tds_read_b32 v61, v60\n\ts_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_mix_max_dsmax6, V8("v_max_i32") "v_mov_b32 v60, 0\n\ts_mov_b32 exec_lo, 0x80008101\n\ts_mov_b32 exec_hi, 0x80008000\n\tds_max_i32 v60, %0\n\ts_mov_b64 exec, -1\n\tds_read_b32 v61, v60\n\ts_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_mix_max_dsmax27x2, V8("v_max_i32") "v_lshrrev_b32 v60, 5, %8\n\tv_and_b32 v60, 4, v60\n\ts_mov_b32 exec_lo, 0x07ffffff\n\ts_mov_b32 exec_hi, 0x07ffffff\n\tds_max_i32 v60, %0\n\ts_mov_b64 exec, -1\n\tds_read_b32 v61, v60\n\ts_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_mix_max_bperm, V8("v_max_i32") "v_and_b32 v60, 0xfc, %7\n\tds_bpermute_b32 v61, v60, %0\n\ts_waitcnt lgkmcnt(0)\n\t")

typedef void (*kern_t)(unsigned *, int);
static double base_rate = 0;
static void run(const char *name, kern_t k, double per_iter, double n_valu, int n_cu, unsigned *out) {
    const int grid = n_cu * 4 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 1024, 0, out, 256);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 1024, 0, out, 256);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)grid * ITER * per_iter;
    const double rate = insts / (ms * 1e-3) / 1e9 / n_cu;  // per CU per ns
    if (base_rate == 0) base_rate = rate;
    // relative to v_add_u32 at 2 cycles per instruction and SIMD; for the mixes: cycles per group of n_valu vector instructions
    const double cyc = 2.0 * base_rate / rate;
    printf("%-22s %8.3f ms  %6.3f per CU per ns  = %5.2f cycles per instruction (v_add_u32 = 2)", name, ms, rate, cyc);
    if (n_valu > 0) printf("; %6.1f cycles per group of %g vector + the rest", cyc * per_iter / (per_iter / (256.0 / 8 * 1)) , n_valu);
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, clockRate %d kHz; 8 wavefronts per SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int n_cu = p.multiProcessorCount;
    unsigned *out;
    CHECK(hipMalloc((void **)&out, (size_t)n_cu * 32 * 64 * sizeof(unsigned)));
#define R(k) run(#k, k, 256, 0, n_cu, out)
    R(k_add); R(k_sub); R(k_and); R(k_or); R(k_xor); R(k_mov); R(k_lshl); R(k_lshr); R(k_ashr); R(k_max_i); R(k_min_u);
    R(k_min3); R(k_max3); R(k_add3); R(k_lshl_add); R(k_add_lshl); R(k_and_or); R(k_lshl_or); R(k_bfe); R(k_alignbit);
    R(k_perm); R(k_mad24); R(k_sad); R(k_ffbl); R(k_add_lit); R(k_add_sgpr); R(k_add_e64); R(k_max_e64);
    R(k_cmp_vcc); R(k_cmp_sgpr); R(k_cmp_eq_const); R(k_cndmask_sgpr);
    R(k_mov_dpp_shr); R(k_mov_dpp_ror); R(k_mov_dpp_row); R(k_max_dpp_row); R(k_max_dpp_bc15); R(k_add_dpp_row);
    R(k_readlane); R(k_writelane); R(k_mbcnt);
    printf("-- mixes: cycles per instruction counts the VECTOR instructions only (8 per group)\n");
    run("k_writelane_m0 (+s_mov)", k_writelane_m0, 256, 0, n_cu, out);
    R(k_mix_max_sadd); R(k_mix_max_2sadd); R(k_mix_add_2sadd); R(k_mix_max_nop); R(k_mix_max_add);
    R(k_mix_max_dsread); R(k_mix_max_dsmax64); R(k_mix_max_dsmax6); R(k_mix_max_dsmax27x2); R(k_mix_max_bperm);
    return 0;
}
