// EXPERIMENT (not part of the product, not linked into libmdbn_hip.so): an f32-grade GEMM on the bf16
// matrix pipe.  Every f32 operand is pre-split into three bf16 pieces a = a1 + a2 + a3 (exact: bf16
// keeps f32's exponent); C = sum over the six piece pairs with i + j <= 4 of A_i B_j^T on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation (dropped pairs are <= 3 * 2^-24 relative).
//   A planes: bf16 [3][M][K] (K contiguous)     B planes: bf16 [3][N][K] (K contiguous)
//   C: f32 [M][N];  M, N multiples of 128, K multiple of 32.
// Same wave-specialised structure as the product GEMM: 4 producer waves copy plane tiles global -> LDS,
// 4 consumer waves (2x2 grid of 64x64 wave tiles) read fragments + MFMA; double-buffered 32-deep slices.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (uint4 went to scratch)

constexpr int BM = 128, BN = 128, KB = 32;
constexpr int ROWB = KB * 2 + 16;                 // bytes per LDS row (64 data + 16 pad): b128 reads conflict-free
constexpr int PLANE_B = 128 * ROWB;               // one 128-row plane tile
constexpr int BUF_B = 6 * PLANE_B;                // A1 A2 A3 B1 B2 B3

__global__ __launch_bounds__(512) void bf16x6_gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nt = K / KB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;

    if (wave >= 4) {                               // ---- producers: 3072 16-byte chunks per slice, 12 per thread
        __builtin_amdgcn_s_setprio(3);
        const int tid = threadIdx.x & 255;
        const int c = tid & 3, r0 = tid >> 2;     // chunk within the 64-byte row slice, row (0..63; + 64)
        u32x4 ra[3][2], rb[3][2];
#define X6_LOAD(SLICE)                                                                                       \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                         \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                   \
            const int row = r0 + 64 * hh;                                                                    \
            ra[pl][hh] = *reinterpret_cast<const u32x4*>(A + pl * planeA + (int64_t)(m0 + row) * K + (SLICE) * KB + 8 * c); \
            rb[pl][hh] = *reinterpret_cast<const u32x4*>(B + pl * planeB + (int64_t)(n0 + row) * K + (SLICE) * KB + 8 * c); \
        }
#define X6_STORE(BUF)                                                                                        \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                         \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                   \
            const int row = r0 + 64 * hh;                                                                    \
            unsigned char* base = smem + (BUF) * BUF_B;                                                      \
            *reinterpret_cast<u32x4*>(base + pl * PLANE_B + row * ROWB + 16 * c) = ra[pl][hh];               \
            *reinterpret_cast<u32x4*>(base + (3 + pl) * PLANE_B + row * ROWB + 16 * c) = rb[pl][hh];         \
        }
        X6_LOAD(0); X6_STORE(0);
        if (nt > 1) { X6_LOAD(1); }
        __syncthreads();
        for (int it = 0; it < nt; ++it) {
            if (it + 1 < nt) { X6_STORE((it + 1) & 1); }
            if (it + 2 < nt) { X6_LOAD(it + 2); }
            __syncthreads();
        }
#undef X6_LOAD
#undef X6_STORE
        return;
    }

    // ---- consumers
    const int lane = threadIdx.x & 63, i = lane & 31, q = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2], lo[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[a][b][e] = 0.f; lo[a][b][e] = 0.f; }

    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const unsigned char* base = smem + (it & 1) * BUF_B;
        bf16x8 af[2][3][2], bf[2][3][2];          // fragment double buffer: step s + 1 loads under step s's MFMAs
#define X6_FRAGS(BUFI, S)                                                                                    \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                         \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                      \
            af[BUFI][pl][a] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE_B + (wm + 32 * a + i) * ROWB + (S) * 32 + q * 16); \
            bf[BUFI][pl][a] = *reinterpret_cast<const bf16x8*>(base + (3 + pl) * PLANE_B + (wn + 32 * a + i) * ROWB + (S) * 32 + q * 16); \
        }
        X6_FRAGS(0, 0);
#pragma unroll
        for (int s = 0; s < KB / 16; ++s) {
            const int cb = s & 1;
            if (s + 1 < KB / 16) { X6_FRAGS(cb ^ 1, s + 1); }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#ifdef X6_TWO_ACC
                    // the five small products go to their own accumulator: an MFMA adds its 16 products to C
                    // with the alignment of the largest addend, so small terms added to a large C lose bits
                    lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][2][a], bf[cb][0][b], lo[a][b], 0, 0, 0);
                    lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][2][b], lo[a][b], 0, 0, 0);
                    lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][1][a], bf[cb][1][b], lo[a][b], 0, 0, 0);
                    lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][1][a], bf[cb][0][b], lo[a][b], 0, 0, 0);
                    lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][1][b], lo[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][0][b], acc[a][b], 0, 0, 0);
#else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][2][a], bf[cb][0][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][2][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][1][a], bf[cb][1][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][1][a], bf[cb][0][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][1][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][0][a], bf[cb][0][b], acc[a][b], 0, 0, 0);
#endif
                }
            // one fragment read of the next step behind every second MFMA
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#undef X6_FRAGS
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                C[(int64_t)(m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * q) * N + n0 + wn + 32 * b + i] = acc[a][b][e] + lo[a][b][e];
}

extern "C" int bf16x6_gemm(void* stream, const void* A, const void* B, float* C, int M, int N, int K)
{
    if (M % BM || N % BN || K % KB) return -1;
    static bool set = false;
    const int lds = 2 * BUF_B;
    if (!set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bf16x6_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
        set = true;
    }
    hipLaunchKernelGGL(bf16x6_gemm_kernel, dim3((M / BM) * (N / BN)), dim3(512), lds, (hipStream_t)stream,
                       (const uint16_t*)A, (const uint16_t*)B, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
