// scripts/ubench/fetch_shapes.hip -- what FETCH_SIZE counts for the read shapes of k_align2:
// every kernel reads a KNOWN number of bytes exactly once from a buffer far larger than the L2s + MALL, in one of the
// shapes the alignment kernel uses, so `rocprofv3 --pmc FETCH_SIZE` on this binary gives the factor between the counter
// (KiB, 64-byte requests) and the bytes that had to come from HBM, per shape:
//   shape_stream16   16 contiguous bytes per lane (k_pack's reads; the tape records of the trace-back)
//   shape_dword_s16  one dword per lane, lanes 16 bytes apart, four such loads per 64 bytes  (the LDS window fill)
//   shape_dwordx2    8 bytes per lane from a window of a few cache lines per wavefront        (a snake beyond 16 bases)
//   shape_gather4    one dword per lane at an address of its own (one 64-byte line per lane)  (the trace-back's cell bytes)
// build: hipcc --offload-arch=gfx950 -O2 fetch_shapes.hip -o fetch_shapes.bin ; run under rocprofv3 --kernel-trace --pmc FETCH_SIZE (round 5: git show 40b807a:scripts/r05_evidence.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
__global__ void shape_stream16(const uint4 *p, u32 *sink, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void shape_dword_s16(const u32 *p, u32 *sink, size_t n_words) {   // all four dwords of every 16 bytes, one load each
    size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4;
    u32 acc = 0;
    for (; i + 3 < n_words; i += (size_t)gridDim.x * blockDim.x * 4)
        for (int j = 0; j < 4; j++) acc ^= __builtin_nontemporal_load(p + i + j);
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void shape_dwordx2(const u32 *p, u32 *sink, size_t n_words) {   // a wavefront reads 8 bytes per lane out of 256 bytes: 2 lanes share
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    const size_t n_wave = (size_t)gridDim.x * blockDim.x / 64;
    u32 acc = 0;
    for (size_t w = wave; (w + 1) * 64 <= n_words; w += n_wave) {
        const uint2 v = *(const uint2 *)(p + w * 64 + (lane >> 1) * 2);
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void shape_gather4(const u32 *p, u32 *sink, size_t n_lines) {   // one dword of every 64-byte line, each line once
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u32 acc = 0;
    for (; i < n_lines; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i * 16 + (i * 7 & 15)];
    if (acc == 0x12345678u) *sink = acc;
}
int main() {
    const size_t bytes = (size_t)4 << 30;   // 4 GiB, read once by each kernel
    u32 *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "no memory\n"); return 1; }
    hipMemset(buf, 0x5a, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 32, block = 256;
    shape_stream16<<<grid, block>>>((const uint4 *)buf, sink, bytes / 16);
    shape_dword_s16<<<grid, block>>>(buf, sink, bytes / 4);
    shape_dwordx2<<<grid, block>>>(buf, sink, bytes / 4);
    shape_gather4<<<grid, block>>>(buf, sink, bytes / 64);
    hipDeviceSynchronize();
    printf("known bytes: shape_stream16 %zu shape_dword_s16 %zu shape_dwordx2 %zu shape_gather4 %zu (lines of 64 B touched: all of %zu bytes)\n",
           bytes, bytes, bytes, bytes / 16, bytes);
    return 0;
}
