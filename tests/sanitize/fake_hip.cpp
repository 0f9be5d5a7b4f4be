// TEST INFRASTRUCTURE (tests/test_sanitize.py): a stand-in for libamdhip64 so that the HOST side of the whole library -- the SRS
// registry, the commitment cache, the deferred rounds, the host pool, the host tails -- can run under AddressSanitizer / UBSan /
// ThreadSanitizer on a machine without a GPU (GPU sanitizers are not available on the pool).  The library is compiled with
// `hipcc --offload-host-only` (kernels become host stubs) and linked against THIS file instead of the HIP runtime:
//   * "device" memory is zero-initialised host memory, copies are memcpy, streams and events are tokens (everything is synchronous);
//   * a kernel launch does nothing.  Device results are therefore all-zero bytes -- the point at infinity, the zero digest -- which
//     is enough for what the sanitizers are here to watch: locks, reference counts, list and buffer management, thread hand-offs.
// Arithmetic on real points is covered by the host-only entry points the harness calls directly (zk_g1_sum_partials*, wire.hip).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

static std::atomic<long> g_launches{0}, g_allocs{0}, g_frees{0};
extern "C" long fake_hip_launches() { return g_launches.load(); }
extern "C" long fake_hip_live_allocations() { return g_allocs.load() - g_frees.load(); }
// A "device" of `budget` bytes (0 = unbounded, reported as 64 GiB free of 256): hipMalloc fails with hipErrorOutOfMemory beyond it and
// hipMemGetInfo reports what is left -- the memory budget of the deferred rounds and the error paths behind a failed allocation
// run under the sanitizers through this.
static std::mutex g_mem_mu;
static std::map<void*, size_t> g_mem;
static size_t g_live_bytes = 0, g_budget = 0, g_report_extra = 0;
// report_extra: hipMemGetInfo over-reports the free memory by this much (fragmentation, another process): the estimate says "fits",
// hipMalloc says no
extern "C" void fake_hip_set_memory(size_t budget, size_t report_extra) {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    g_budget = budget;
    g_report_extra = report_extra;
}
extern "C" size_t fake_hip_live_bytes() {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    return g_live_bytes;
}

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake hip"; }
hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mem_mu);
        if (g_budget && g_live_bytes + n > g_budget) return hipErrorOutOfMemory;
    }
    *p = calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    ++g_allocs;
    std::lock_guard<std::mutex> lk(g_mem_mu);
    g_mem[*p] = n;
    g_live_bytes += n;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (p) {
        ++g_frees;
        std::lock_guard<std::mutex> lk(g_mem_mu);
        auto it = g_mem.find(p);
        if (it != g_mem.end()) {
            g_live_bytes -= it->second;
            g_mem.erase(it);
        }
    }
    free(p);
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* fr, size_t* tot) {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    if (g_budget) {
        *tot = g_budget;
        *fr = (g_budget > g_live_bytes ? g_budget - g_live_bytes : 0) + g_report_extra;
    } else {
        *tot = (size_t)256 << 30;
        *fr = (size_t)64 << 30;
    }
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) {
    memset(a, 0, sizeof *a);
    return hipErrorInvalidValue;      // "not a registered pointer": the library then treats the buffer as pageable
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    if (n) memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(d, s, n, k, nullptr); }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (n) memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = (hipStream_t)calloc(1, 8);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    free((void*)s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = (hipEvent_t)calloc(1, 8);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    free((void*)e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// what the module constructor of a HIP object file calls (registration of its kernels with the runtime)
void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
// what a kernel's host stub calls
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
    *g = dim3(1);
    *b = dim3(1);
    *sh = 0;
    *st = nullptr;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) {
    ++g_launches;
    return hipSuccess;
}
}
