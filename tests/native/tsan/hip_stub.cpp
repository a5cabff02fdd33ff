// TEST infrastructure (never shipped): a do-nothing HIP runtime for the ThreadSanitizer run of the plugin's HOST logic — rendezvous of
// decoding states, lanes, stream / event ordering, upload ring, logits mirror (tests/test_host.py::test_plugin_host_threading_under_tsan).
// "Device" memory is host memory, copies are memcpy, streams and events are inert handles, everything completes at once.
// Reference practice: the sanitizer jobs of the reference's CI (.github/workflows/build-sanitize.yml:38).
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipGetDeviceCount(int * n) { *n = 2; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char * hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t * p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "stub gfx950"); strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t) 1 << 38; return hipSuccess; }
hipError_t hipMemGetInfo(size_t * f, size_t * t) { *f = (size_t) 1 << 37; *t = (size_t) 1 << 38; return hipSuccess; }
hipError_t hipMalloc(void ** p, size_t n) { *p = nullptr; if (posix_memalign(p, 4096, n ? n : 1) != 0) return hipErrorOutOfMemory; memset(*p, 0, n ? n : 1); return hipSuccess; }      // (device allocations are at least 256-byte aligned: ggml-alloc relies on the buffer type's alignment)
hipError_t hipFree(void * p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void ** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void * p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void ** d, void * h, unsigned int) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void * d, const void * s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void * d, const void * s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void * d, int, const void * s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void * d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void * d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t * s, unsigned int) { *s = (hipStream_t) calloc(1, 16); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t * e) { *e = (hipEvent_t) calloc(1, 16); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t * e, unsigned) { *e = (hipEvent_t) calloc(1, 16); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float * ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int * can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }
}
