// Stand-in for a collective's kernel on a ONE-GPU box: R workgroups with the footprint of RCCL's gfx950 all-reduce
// kernel (librccl.so 7.2.0, ncclDevKernel_Generic: 248-256 VGPRs, 37,664 B LDS, up to 512 threads per workgroup --
// read from the code object's notes) that hold their CUs for a given time while streaming a
// buffer.  It moves no data between GPUs and proves nothing about xGMI; it shows what the step pays when a kernel of
// that shape runs beside it.  scripts/dp_contention_probe.py drives it.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(512) void occupier_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4,
                                                      int64_t ticks)     // 100 MHz wall-clock ticks
{
    __shared__ volatile float lds[37664 / 4];
    asm volatile("v_mov_b32 v250, 0" ::: "v250");                        // the register footprint: 256 VGPRs per wave
    lds[threadIdx.x] = 0.f;
    const int64_t t0 = wall_clock64();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // every wave leaves when the time is up: the loop is bounded by the clock, not by the data
    while (wall_clock64() - t0 < ticks) {
        const float4 v = src[i % n4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        dst[i % n4] = acc;
        i += stride;
        // ~32 bytes per thread per microsecond: 32 x 512 threads move ~0.5 TB/s, about what an 8-rank ring all-reduce of
        // the 16.8 MB statistics buffer reads and writes locally in 120 us
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc.x == 12345.678f) dst[0] = make_float4(lds[(threadIdx.x * 7) % (37664 / 4)], 0.f, 0.f, 0.f);   // keep both alive
}

extern "C" int occupier_launch(void* stream, int blocks, int threads, const void* src, void* dst, long long n_bytes,
                               double microseconds)
{
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(occupier_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                       (int64_t)(n_bytes / 16), (int64_t)(microseconds * 100.0));
    return (int)hipGetLastError();
}

// Fork / join with the library's own events, to price the cross-stream dependency itself:
//   flags = hipEventDisableTiming (0x2), optionally | hipEventReleaseToDevice (0x40000000) / hipEventDisableSystemFence (0x20000000)
static hipEvent_t g_fork = nullptr, g_join[2] = {nullptr, nullptr};
extern "C" int occupier_events(unsigned flags)
{
    for (hipEvent_t* e : {&g_fork, &g_join[0], &g_join[1]}) {
        if (*e) (void)hipEventDestroy(*e);
        hipError_t rc = hipEventCreateWithFlags(e, flags);
        if (rc != hipSuccess) return (int)rc;
    }
    return 0;
}
extern "C" int occupier_fork(void* compute, void* side, int slot, int blocks, int threads, const void* src, void* dst, long long n_bytes,
                             double microseconds)
{
    hipError_t rc = hipEventRecord(g_fork, (hipStream_t)compute);
    if (rc != hipSuccess) return (int)rc;
    rc = hipStreamWaitEvent((hipStream_t)side, g_fork, 0);
    if (rc != hipSuccess) return (int)rc;
    int r = occupier_launch(side, blocks, threads, src, dst, n_bytes, microseconds);
    if (r) return r;
    return (int)hipEventRecord(g_join[slot & 1], (hipStream_t)side);
}
extern "C" int occupier_join(void* compute, int slot) { return (int)hipStreamWaitEvent((hipStream_t)compute, g_join[slot & 1], 0); }

// The same fork / join through STREAM MEMORY OPERATIONS instead of events: the producer stream writes a sequence number
// to signal memory when its preceding work is done (hipStreamWriteValue32), the consumer stream's command processor
// polls it (hipStreamWaitValue32, >=) -- no inter-queue signal machinery.  Priced against the event forms by
// scripts/dp_contention_probe.py --events value.
static uint32_t* g_sig_fork = nullptr;
static uint32_t* g_sig_join = nullptr;
extern "C" int occupier_values_init(void)
{
    int ok = 0;
    hipError_t rc = hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, 0);
    if (rc != hipSuccess) return (int)rc;
    if (!ok) return -1;
    for (uint32_t** p : {&g_sig_fork, &g_sig_join}) {
        rc = hipExtMallocWithFlags(reinterpret_cast<void**>(p), 8, hipMallocSignalMemory);
        if (rc != hipSuccess) return (int)rc;
        rc = hipMemset(*p, 0, 8);
        if (rc != hipSuccess) return (int)rc;
    }
    return 0;
}
extern "C" int occupier_fork_value(void* compute, void* side, unsigned seq, int blocks, int threads, const void* src, void* dst,
                                   long long n_bytes, double microseconds)
{
    hipError_t rc = hipStreamWriteValue32((hipStream_t)compute, g_sig_fork, seq, 0);
    if (rc != hipSuccess) return (int)rc;
    rc = hipStreamWaitValue32((hipStream_t)side, g_sig_fork, seq, hipStreamWaitValueGte, 0xffffffffu);
    if (rc != hipSuccess) return (int)rc;
    int r = occupier_launch(side, blocks, threads, src, dst, n_bytes, microseconds);
    if (r) return r;
    return (int)hipStreamWriteValue32((hipStream_t)side, g_sig_join, seq, 0);
}
extern "C" int occupier_join_value(void* compute, unsigned seq)
{
    return (int)hipStreamWaitValue32((hipStream_t)compute, g_sig_join, seq, hipStreamWaitValueGte, 0xffffffffu);
}
