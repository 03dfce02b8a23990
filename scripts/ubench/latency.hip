// latency.hip -- dependent-chain latencies of the cross-lane primitives the MSA kernels lean on
// (one wavefront per SIMD, nothing else running): cycles per operation, s_memtime clock.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/latency.bin scripts/ubench/latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 4000
#define REP8(s) s s s s s s s s

#define CHAIN_KERNEL(name, decl, body, result)                                             \
    __global__ __launch_bounds__(64) void name(unsigned *out, unsigned long long *clk) {   \
        __shared__ unsigned lds[256];                                                      \
        lds[threadIdx.x] = (threadIdx.x * 4 + 4) & 255; lds[threadIdx.x + 64] = 0;          \
        __syncthreads();                                                                    \
        decl;                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                         \
        for (int i = 0; i < ITER; i++) { REP8(body) }                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                         \
        out[blockIdx.x * 64 + threadIdx.x] = (unsigned)(result) + lds[0];                   \
        if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;                                    \
    }

CHAIN_KERNEL(k_bpermute, unsigned a = (threadIdx.x * 4 + 4) & 255,
             a = (unsigned)__builtin_amdgcn_ds_bpermute((int)a, (int)a);, a)
CHAIN_KERNEL(k_permute, unsigned a = (threadIdx.x * 4 + 4) & 255,
             a = (unsigned)__builtin_amdgcn_ds_permute((int)a, (int)a);, a)
CHAIN_KERNEL(k_ds_read, unsigned a = threadIdx.x * 4,
             a = *(volatile unsigned *)((char *)lds + (a & 255));, a)
CHAIN_KERNEL(k_v_add, unsigned a = threadIdx.x,
             asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));, a)
CHAIN_KERNEL(k_v_max, unsigned a = threadIdx.x,
             asm volatile("v_max_u32 %0, %0, %0" : "+v"(a));, a)
CHAIN_KERNEL(k_dpp_max, unsigned a = threadIdx.x,
             asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));, a)
CHAIN_KERNEL(k_readlane_rt, unsigned a = threadIdx.x; unsigned s = 0,
             asm volatile("v_readlane_b32 %1, %0, 3\n\tv_add_u32 %0, %1, %0" : "+v"(a), "+s"(s));, a)
CHAIN_KERNEL(k_cmp_cnd, unsigned a = threadIdx.x; unsigned long long m = 0,
             asm volatile("v_cmp_lt_u32 %1, 5, %0\n\tv_cndmask_b32 %0, %0, %0, %1" : "+v"(a), "+s"(m));, a)
CHAIN_KERNEL(k_cnd_vcc, unsigned a = threadIdx.x,
             asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a) : : "vcc");, a)
CHAIN_KERNEL(k_cnd_sgpr, unsigned a = threadIdx.x; unsigned long long m = 0x5555555555555555ull,
             asm volatile("v_cndmask_b32 %0, %0, %0, %1" : "+v"(a) : "s"(m));, a)
CHAIN_KERNEL(k_lshl_add, unsigned a = threadIdx.x,
             asm volatile("v_lshl_add_u32 %0, %0, 1, %0" : "+v"(a));, a)
CHAIN_KERNEL(k_s_add, unsigned a = blockIdx.x,
             asm volatile("s_add_u32 %0, %0, %0" : "+s"(a) : : "scc");, a)

typedef void (*kern_t)(unsigned *, unsigned long long *);
static void run(const char *name, kern_t k, int ops_per_body, int waves_per_cu, int n_cu, unsigned *out, unsigned long long *clk) {
    const int grid = n_cu * waves_per_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, clk);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), clk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double ticks = 0; for (auto t : h) ticks += (double)t; ticks /= grid;
    const double n = (double)ITER * 8 * ops_per_body;
    printf("%-22s %2d waves/CU  %8.3f ms  %7.1f ns/op = %6.1f cycles at 2.4 GHz  (s_memtime: %.1f ticks/op)\n",
           name, waves_per_cu, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4, ticks / n);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    unsigned *out; unsigned long long *clk;
    CHECK(hipMalloc((void **)&out, (size_t)n_cu * 32 * 64 * 4));
    CHECK(hipMalloc((void **)&clk, (size_t)n_cu * 32 * 8));
    for (int w : {1, 4, 12}) {
        run("ds_bpermute chain", k_bpermute, 1, w, n_cu, out, clk);
        run("ds_permute chain", k_permute, 1, w, n_cu, out, clk);
        run("ds_read_b32 chain", k_ds_read, 1, w, n_cu, out, clk);
    }
    run("v_add_u32 chain", k_v_add, 1, 1, n_cu, out, clk);
    run("v_max_u32 chain", k_v_max, 1, 1, n_cu, out, clk);
    run("s_nop1+v_max_dpp chain", k_dpp_max, 1, 1, n_cu, out, clk);
    run("readlane->v_add chain", k_readlane_rt, 2, 1, n_cu, out, clk);
    run("v_cmp->v_cndmask chain", k_cmp_cnd, 2, 1, n_cu, out, clk);
    run("v_cndmask(vcc) chain", k_cnd_vcc, 1, 1, n_cu, out, clk);
    run("v_cndmask(sgpr) chain", k_cnd_sgpr, 1, 1, n_cu, out, clk);
    run("v_lshl_add chain", k_lshl_add, 1, 1, n_cu, out, clk);
    run("s_add_u32 chain", k_s_add, 1, 1, n_cu, out, clk);
    return 0;
}
