// tests/emu/simt/hip/hip_runtime.h -- TEST INFRASTRUCTURE: stands in for <hip/hip_runtime.h> when
// kernel sources are compiled for the host-side SIMT emulator (simt.h); g++ finds this directory
// first on its include path.  Only what falcon_amd/csrc/fa_internal.h and the MSA kernels name.
#pragma once
#include "../simt.h"
typedef void *hipStream_t;
typedef void *hipEvent_t;
