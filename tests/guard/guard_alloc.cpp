// Test infrastructure (not product code): device buffers with UNMAPPED guard pages on both sides.
//
// wt_guard_alloc(bytes, at_end) reserves a virtual range of [guard | payload pages | guard], maps physical memory
// behind the payload pages only (HIP virtual memory management), and returns a pointer such that the buffer either
// ENDS exactly at the last mapped byte (at_end = 1: a one-element overrun faults) or STARTS at the first mapped byte
// (at_end = 0: a one-element underrun faults).  tests/test_gpu_guard.py runs every kernel of libwtalign.so on such
// buffers.  Built by __graft_entry__.build() into tests/guard/libwtguard.so (host-only code, links libamdhip64).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct Guarded {
    void *base;
    size_t reserved, mapped, granule;
    hipMemGenericAllocationHandle_t handle;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s", #x, hipGetErrorString(e_)); return nullptr; } } while (0)
static char g_err[256] = "";

extern "C" const char *wt_guard_error(void) { return g_err; }

extern "C" void *wt_guard_alloc(size_t bytes, int at_end, void **ticket) {
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t g = 0;
    CK(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum));
    if (bytes == 0) bytes = 1;
    const size_t mapped = (bytes + g - 1) / g * g;
    Guarded *t = new Guarded();
    t->granule = g;
    t->mapped = mapped;
    t->reserved = mapped + 2 * g;
    CK(hipMemAddressReserve(&t->base, t->reserved, g, nullptr, 0));
    CK(hipMemCreate(&t->handle, mapped, &prop, 0));
    char *payload = (char *)t->base + g;
    CK(hipMemMap(payload, mapped, 0, t->handle, 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(payload, mapped, &acc, 1));
    *ticket = t;
    return at_end ? payload + (mapped - bytes) : payload;
}

extern "C" int wt_guard_free(void *ticket) {
    Guarded *t = (Guarded *)ticket;
    if (!t) return 0;
    char *payload = (char *)t->base + t->granule;
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(payload, t->mapped);
    (void)hipMemRelease(t->handle);
    (void)hipMemAddressFree(t->base, t->reserved);
    delete t;
    return 0;
}
