// A HOST stand-in for <hip/hip_runtime.h>, for ONE purpose: compiling basebandboard_amd/csrc/bbb_api.hip -- the library's host
// scheduler, unchanged -- with g++ and running it against a model of streams and events that checks every buffer access for
// ordering (tests/sched_model/model.cpp, tests/test_sched_model.py).  Nothing here computes anything.
#pragma once
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct MockStream *hipStream_t;
typedef struct MockEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

extern "C" {
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t s);
hipError_t hipFreeAsync(void *p, hipStream_t s);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
}
// (the typed forms of the real header)
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc(reinterpret_cast<void **>(p), n); }
template <class T> static inline hipError_t hipMallocAsync(T **p, size_t n, hipStream_t s) { return hipMallocAsync(reinterpret_cast<void **>(p), n, s); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags) { return hipHostMalloc(reinterpret_cast<void **>(p), n, flags); }
