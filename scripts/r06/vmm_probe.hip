// vmm_probe.hip — experiment only (VERDICT round 5, next #2b/#2c): iterates mapped through the HIP virtual-memory API
// (hipMemCreate / hipMemAddressReserve / hipMemMap) with a chosen physical chunk size and virtual alignment, to see whether the
// page-table fragment size the driver ends up with changes the SpMM's address-translation behaviour (UTCL1 misses at 100 GB
// iterates) or explains the 10-12 % "placement classes" of plain hipMalloc pairs.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/r06/vmm_probe.hip -o gpurun_out/libvmm_probe.so
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

extern "C" int vmm_granularity(int device, size_t *gmin, size_t *grec) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (hipMemGetAllocationGranularity(gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return -1;
    if (hipMemGetAllocationGranularity(grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return -2;
    return 0;
}

// Reserves `bytes` (rounded up to `chunk`) of virtual space aligned to `va_align` and backs it chunk by chunk.
extern "C" int vmm_alloc(int device, size_t bytes, size_t va_align, size_t chunk, void **out, size_t *mapped) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (chunk == 0) chunk = bytes;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return -1;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void *ptr = nullptr;
    hipError_t e = hipMemAddressReserve(&ptr, total, va_align, nullptr, 0);
    if (e != hipSuccess) { fprintf(stderr, "reserve: %s\n", hipGetErrorString(e)); return -2; }
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) { fprintf(stderr, "create at %zu: %s\n", off, hipGetErrorString(e)); return -3; }
        e = hipMemMap((char *)ptr + off, chunk, 0, h, 0);
        if (e != hipSuccess) { fprintf(stderr, "map at %zu: %s\n", off, hipGetErrorString(e)); return -4; }
        (void)hipMemRelease(h);    // the mapping keeps the physical memory alive
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(ptr, total, &acc, 1);
    if (e != hipSuccess) { fprintf(stderr, "access: %s\n", hipGetErrorString(e)); return -5; }
    *out = ptr;
    *mapped = total;
    return 0;
}

extern "C" int vmm_free(void *ptr, size_t mapped) {
    if (hipMemUnmap(ptr, mapped) != hipSuccess) return -1;
    if (hipMemAddressFree(ptr, mapped) != hipSuccess) return -2;
    return 0;
}

extern "C" int plain_alloc(size_t bytes, void **out) { return hipMalloc(out, bytes) == hipSuccess ? 0 : -1; }
extern "C" int plain_free(void *p) { return hipFree(p) == hipSuccess ? 0 : -1; }
