// issue_rates.hip -- what one gfx950 CU issues per clock, measured: the denominators of the
// issue-bound table in DESIGN.md section 4 (k_align is bound by instruction issue, not HBM).
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/issue_rates.bin scripts/ubench/issue_rates.hip
//   scripts/ubench/issue_rates.bin
//
// Every kernel runs 8 wavefronts per SIMD (8192 one-wave workgroups on 256 CUs), each wave
// ITER x 256 instructions of one kind on 8 independent registers (no dependent chain
// shorter than 8 instructions), and reports wave-instructions per second for the chip and
// per CU and clock (clock measured with s_memtime against the HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(s) s s s s s s s s
#define REP32(s) REP8(s) REP8(s) REP8(s) REP8(s)

#define V8(op) \
    op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
    op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8\n\t"

#define VALU_KERNEL(name, body)                                                           \
    __global__ __launch_bounds__(64, 8) void name(unsigned *out, unsigned long long *clk) { \
        unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5,    \
                 g = a + 6, h = a + 7, k = blockIdx.x | 1;                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                         \
        for (int i = 0; i < ITER; i++)                                                      \
            asm volatile(REP32(body)                                                        \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) \
                         : "v"(k) : "vcc");                                                 \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                         \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;                 \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                    \
    }

VALU_KERNEL(k_v_add, V8("v_add_u32"))
VALU_KERNEL(k_v_and, V8("v_and_b32"))
VALU_KERNEL(k_v_max, V8("v_max_i32"))
VALU_KERNEL(k_v_lshl, V8("v_lshlrev_b32"))
#define A8 \
    "v_alignbit_b32 %0, %0, %1, %8\n\t" "v_alignbit_b32 %1, %1, %2, %8\n\t" "v_alignbit_b32 %2, %2, %3, %8\n\t" \
    "v_alignbit_b32 %3, %3, %4, %8\n\t" "v_alignbit_b32 %4, %4, %5, %8\n\t" "v_alignbit_b32 %5, %5, %6, %8\n\t" \
    "v_alignbit_b32 %6, %6, %7, %8\n\t" "v_alignbit_b32 %7, %7, %0, %8\n\t"
VALU_KERNEL(k_v_alignbit, A8)
#define C8 \
    "v_cndmask_b32 %0, %0, %8, vcc\n\t" "v_cndmask_b32 %1, %1, %8, vcc\n\t" "v_cndmask_b32 %2, %2, %8, vcc\n\t" \
    "v_cndmask_b32 %3, %3, %8, vcc\n\t" "v_cndmask_b32 %4, %4, %8, vcc\n\t" "v_cndmask_b32 %5, %5, %8, vcc\n\t" \
    "v_cndmask_b32 %6, %6, %8, vcc\n\t" "v_cndmask_b32 %7, %7, %8, vcc\n\t"
VALU_KERNEL(k_v_cndmask, C8)
#define D8 \
    "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_max_i32_dpp %1, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_i32_dpp %2, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_max_i32_dpp %3, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_i32_dpp %4, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_max_i32_dpp %5, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_i32_dpp %6, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_max_i32_dpp %7, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
VALU_KERNEL(k_v_dpp, D8)
#define M8 \
    "v_cmp_lt_i32 vcc, %0, %8\n\t" "v_cmp_lt_i32 vcc, %1, %8\n\t" "v_cmp_lt_i32 vcc, %2, %8\n\t" "v_cmp_lt_i32 vcc, %3, %8\n\t" \
    "v_cmp_lt_i32 vcc, %4, %8\n\t" "v_cmp_lt_i32 vcc, %5, %8\n\t" "v_cmp_lt_i32 vcc, %6, %8\n\t" "v_cmp_lt_i32 vcc, %7, %8\n\t"
VALU_KERNEL(k_v_cmp, M8)
#define F8 V8("v_fma_f32")
__global__ __launch_bounds__(64, 8) void k_v_fma(unsigned *out, unsigned long long *clk) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, k = 1.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++)
        asm volatile(REP32("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                           "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a + b + c + d + e + f + g + h);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// scalar: 8 independent SGPRs, s_add_u32 / s_and_b64 mix
#define S8 \
    "s_add_u32 %0, %0, %8\n\t" "s_add_u32 %1, %1, %8\n\t" "s_add_u32 %2, %2, %8\n\t" "s_add_u32 %3, %3, %8\n\t" \
    "s_add_u32 %4, %4, %8\n\t" "s_add_u32 %5, %5, %8\n\t" "s_add_u32 %6, %6, %8\n\t" "s_add_u32 %7, %7, %8\n\t"
#define S8B \
    "s_and_b64 %0, %0, %1\n\t" "s_or_b64 %1, %1, %2\n\t" "s_and_b64 %2, %2, %3\n\t" "s_or_b64 %3, %3, %0\n\t" \
    "s_and_b64 %0, %0, %1\n\t" "s_or_b64 %1, %1, %2\n\t" "s_and_b64 %2, %2, %3\n\t" "s_or_b64 %3, %3, %0\n\t"
__global__ __launch_bounds__(64, 8) void k_s_add(unsigned *out, unsigned long long *clk) {
    unsigned a = blockIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, k = blockIdx.x | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++)
        asm volatile(REP32(S8) : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : "s"(k) : "scc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(64, 8) void k_s_b64(unsigned *out, unsigned long long *clk) {
    unsigned long long a = blockIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = ~a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++)
        asm volatile(REP32(S8B) : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(a + b + c + d);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
// k_align's mix: one scalar instruction beside every vector one (different pipes: do they overlap?)
#define VS8 \
    "v_add_u32 %0, %0, %12\n\t" "s_add_u32 %8, %8, %13\n\t" "v_add_u32 %1, %1, %12\n\t" "s_add_u32 %9, %9, %13\n\t" \
    "v_add_u32 %2, %2, %12\n\t" "s_add_u32 %10, %10, %13\n\t" "v_add_u32 %3, %3, %12\n\t" "s_add_u32 %11, %11, %13\n\t" \
    "v_add_u32 %4, %4, %12\n\t" "s_add_u32 %8, %8, %13\n\t" "v_add_u32 %5, %5, %12\n\t" "s_add_u32 %9, %9, %13\n\t" \
    "v_add_u32 %6, %6, %12\n\t" "s_add_u32 %10, %10, %13\n\t" "v_add_u32 %7, %7, %12\n\t" "s_add_u32 %11, %11, %13\n\t"
__global__ __launch_bounds__(64, 8) void k_mix(unsigned *out, unsigned long long *clk) {
    unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7, k = blockIdx.x | 1;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, sk = blockIdx.x | 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++)
        asm volatile(REP32(VS8)
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h),
                       "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                     : "v"(k), "s"(sk) : "scc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(unsigned *, unsigned long long *);

static void run(const char *name, kern_t k, double per_iter, int waves_per_simd, int n_cu, unsigned *out,
                unsigned long long *clk) {
    const int grid = n_cu * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), clk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double ticks = 0;
    for (auto t : h) ticks += (double)t;
    ticks /= grid;
    const double insts = (double)grid * ITER * per_iter;
    const double rate = insts / (ms * 1e-3);
    printf("%-14s %2d waves/SIMD  %8.3f ms  %7.1f G wave-instr/s  = %5.3f per CU per ns; "
           "s_memtime ticks per wave %.3g (%.1f MHz if the wave ran the whole launch)\n",
           name, waves_per_simd, ms, rate / 1e9, rate / 1e9 / n_cu, ticks, ticks / (ms * 1e3));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int n_cu = p.multiProcessorCount;
    unsigned *out;
    unsigned long long *clk;
    CHECK(hipMalloc((void **)&out, (size_t)n_cu * 32 * 64 * sizeof(unsigned)));
    CHECK(hipMalloc((void **)&clk, (size_t)n_cu * 32 * sizeof(unsigned long long)));
    for (int w : {8, 4, 2, 1}) {
        run("v_add_u32", k_v_add, 256, w, n_cu, out, clk);
        run("s_add_u32", k_s_add, 256, w, n_cu, out, clk);
    }
    run("v_and_b32", k_v_and, 256, 8, n_cu, out, clk);
    run("v_max_i32", k_v_max, 256, 8, n_cu, out, clk);
    run("v_lshlrev_b32", k_v_lshl, 256, 8, n_cu, out, clk);
    run("v_alignbit", k_v_alignbit, 256, 8, n_cu, out, clk);
    run("v_cndmask", k_v_cndmask, 256, 8, n_cu, out, clk);
    run("v_max_dpp", k_v_dpp, 256, 8, n_cu, out, clk);
    run("v_cmp", k_v_cmp, 256, 8, n_cu, out, clk);
    run("v_fma_f32", k_v_fma, 256, 8, n_cu, out, clk);
    run("s_and/or_b64", k_s_b64, 256, 8, n_cu, out, clk);
    run("v_add+s_add", k_mix, 512, 8, n_cu, out, clk);
    return 0;
}
