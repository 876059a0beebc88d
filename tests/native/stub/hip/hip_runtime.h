/* Host stand-in for <hip/hip_runtime.h>, for the CPU sanitizer build of the library's HOST code only (tests/native/Makefile:
 * csrc/planner.cpp and the handle / workspace-layout half of csrc/program.hip under -fsanitize=address,undefined).  "Device"
 * memory is host memory, so AddressSanitizer sees every buffer the handles carve out of a caller's workspace; the kernels are
 * replaced by kernel_stubs.cpp, which touch the extents of their arguments.  Test infrastructure -- never part of libvd_hip.so. */
#ifndef VD_TEST_HIP_STUB_H
#define VD_TEST_HIP_STUB_H
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 } hipError_t;
typedef struct vdStubStream* hipStream_t;
typedef enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;

static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
#endif
