// tests/san/stub/hip/hip_runtime.h -- TEST INFRASTRUCTURE: a host stand-in for the part of the HIP runtime API that
// falcon_amd/csrc/engine.hip (pure host code) uses, so that the batch engine's threading -- submitting threads, the
// planner thread, batches freed while in flight, the block cache -- can run under ThreadSanitizer, which cannot load the
// real runtime (tests/san/engine_san.cpp, `make -C falcon_amd/csrc tsan_engine`).  "Device" memory is host memory, a
// stream executes what is put on it at once, an event is a time stamp.  Never part of the product.
#pragma once
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdint.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct stub_stream *hipStream_t;
typedef struct stub_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; char name[64]; };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "stub error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 2; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p)); p->multiProcessorCount = 4; p->totalGlobalMem = (size_t)1 << 32; return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)1 << 31; *t = (size_t)1 << 32; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *l, int *g) { *l = 0; *g = -1; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(stub_event)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
// (events are written by the thread that "records" them and read by the one that asks for a time: relaxed atomics,
// so that the sanitizer reports the engine's races and not the stub's)
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    __atomic_store(&e->t, &t, __ATOMIC_RELEASE);
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    double ta, tb;
    __atomic_load(&a->t, &ta, __ATOMIC_ACQUIRE); __atomic_load(&b->t, &tb, __ATOMIC_ACQUIRE);
    *ms = (float)(tb - ta);
    return hipSuccess;
}
