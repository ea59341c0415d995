// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: which (lane, register) of D receives A[la] * B[lb], without and with the
// A-block broadcast (cbsz = 4, abid = q).  Build + run:  hipcc -O2 --offload-arch=gfx950 mfma4x4_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(int* out)      // out[(la * 64 + lb) * 2 + {0,1}] = lane * 4 + e of the (single) non-zero output, count
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const float a = lane == la ? 1.f : 0.f, b = lane == lb ? 1.f : 0.f;
            f4 d = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
            if (MODE == 1) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 4, 0, 0);
            if (MODE == 2) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 4, 5, 0);
            if (MODE == 3) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 4, 15, 0);
            for (int e = 0; e < 4; ++e)
                if (d[e] != 0.f) { atomicAdd(&out[(la * 64 + lb) * 2 + 1], 1); out[(la * 64 + lb) * 2] = lane * 4 + e; }
        }
}

template <int MODE>
static void run(const char* name)
{
    int* d; hipMalloc(&d, 64 * 64 * 2 * sizeof(int)); hipMemset(d, 0, 64 * 64 * 2 * sizeof(int));
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    int bad = 0, hits = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int cnt = h[(la * 64 + lb) * 2 + 1], where = h[(la * 64 + lb) * 2];
            if (!cnt) continue;
            ++hits;
            const int q = MODE == 2 ? 5 : MODE == 3 ? 15 : 0;
            // expectation: plain: block(la) == block(lb), D lane = 4 block + lb % 4, e = la % 4
            //              broadcast: block(la) == q, D lane = lb, e = la % 4
            const bool expect_hit = MODE == 0 ? (la / 4 == lb / 4) : (la / 4 == q);
            const int expect_where = lb * 4 + la % 4;
            if (!expect_hit || cnt != 1 || where != expect_where) {
                if (bad++ < 12) printf("  la %2d lb %2d: count %d at lane %d e %d (expected %s lane %d e %d)\n", la, lb, cnt, where / 4, where % 4,
                                       expect_hit ? "hit" : "NO hit", expect_where / 4, expect_where % 4);
            }
        }
    printf("  %d products landed, %d not where expected (expected %d products)\n", hits, bad, MODE == 0 ? 64 * 4 : 4 * 64);
    hipFree(d);
}

// issue cost: 256 MFMAs on NACC independent accumulator chains, one wave / two waves per SIMD (8 waves of a 512-thread block)
template <int NACC>
__global__ void chain(float* out, long long* cyc)
{
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x, b = 1.0f / (1 + threadIdx.x);
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256 / 16; ++it) {
#define ONE(U) acc[(U) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[(U) % NACC], 4, U, 0);
        ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
#undef ONE
    }
    const long long t1 = clock64();
    f4 s = acc[0];
    for (int j = 1; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
static void time_chain(int threads)
{
    float* o; long long* c; hipMalloc(&o, 4096 * 4); hipMalloc(&c, 64);
    hipLaunchKernelGGL(chain<NACC>, dim3(1), dim3(threads), 0, 0, o, c);
    hipLaunchKernelGGL(chain<NACC>, dim3(1), dim3(threads), 0, 0, o, c);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("  %d chains, %d waves: %lld shader-clock ticks for 256 MFMAs per wave (%.1f per MFMA)\n", NACC, threads / 64, h, h / 256.0);
    hipFree(o); hipFree(c);
}

int main()
{
    printf("== issue cost (clock64 ticks: 100 MHz s_memtime units if the count looks tiny)\n");
    time_chain<1>(64); time_chain<2>(64); time_chain<4>(64); time_chain<1>(512); time_chain<2>(512); time_chain<4>(512);
    run<0>("plain (cbsz 0)");
    run<1>("cbsz 4 abid 0");
    run<2>("cbsz 4 abid 5");
    run<3>("cbsz 4 abid 15");
    return 0;
}
