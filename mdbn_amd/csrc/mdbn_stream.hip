// Register-streaming GEMM on the bf16 matrix pipe at f32 accuracy -- the kernel of the MID-SIZE layers (BASELINE configs
// 4 / 5: 1024 -> 256, 2048 -> 400 at B = 512; reference shapes src/AMLsm2.py:242-251, src/MDBN.py:31-35).
//
// Those passes are too small for 128 x 128 tiles: 16 - 64 tiles need an 4 - 16-way split-K to occupy the chip, every
// pass then is GEMM + [splits][M][N] slabs + a second launch that sums them and applies the activation (20 - 24 us for
// 0.8 GFLOP).  Here the output is cut into 32 x 32 (or 64 x 32) tiles -- hundreds of them, one workgroup each, NO split-K
// across workgroups -- and the whole reduction range streams through the tile's 8 waves straight from L2 into the MFMA
// operand registers (no LDS staging: at this size every operand is L2 / Infinity-Cache resident and the chip has more CUs
// than the problem has 128-wide tiles).  The f32 operands are split into their three exact bf16 pieces IN REGISTERS
// (mdbn_bf16x3.h: the products and order of gemm_bf16x6_kernel -- six piece products, three when the row operand holds
// 0/1 samples), the 8 partial tiles are reduced through LDS in wave order and the launch's epilogue runs on the tile
// (skinny_tile_epilogue: activation + sampling + bias statistics, or the parameter update for the statistics GEMM): ONE
// launch per pass.
//   LAY_K  operand X[rows][ld], K contiguous : lane (i, h) <- X[r0 + i][k16 + 8 h .. + 7]      (two float4)
//   LAY_MN operand X[K][ld], rows contiguous : lane (i, h) <- X[k16 + 8 h + e][r0 + i], e < 8
// Wave w takes the 16-deep steps w, w + 8, ...; three named register sets keep two steps of loads in flight behind the
// one being multiplied (a register array indexed by the step's residue would serialise the loads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_skinny.h"

namespace mdbn {

// eight K-consecutive values of a lane's fragment; every call issues the SAME instructions (the pipelined loop below must
// not contain loads under data-dependent branches: hipcc's wait-count bookkeeping then assumes the fewest loads in flight
// at every join and waits for ALL of them before the first MFMA)
template <int LAY>
__device__ __forceinline__ void stream_load8(const float* p, int64_t ld, int k, float (&f)[8])
{
    if (LAY == LAY_K) {
        const float4 u = *reinterpret_cast<const float4*>(p + k), v = *reinterpret_cast<const float4*>(p + k + 4);
        f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
    } else {
        const float* q = p + (int64_t)k * ld;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = q[(int64_t)e * ld];
    }
}

// the K tail (the last, partial step of a range): loads at clamped addresses, zero on use (the pieces of 0 are 0)
template <int LAY>
__device__ __forceinline__ void stream_load8_tail(const float* p, int64_t ld, int k, int kend, float (&f)[8])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = min(k + e, kend - 1);
        const float v = LAY == LAY_K ? p[kk] : p[(int64_t)kk * ld];
        f[e] = k + e < kend ? v : 0.f;
    }
}

#ifdef MDBN_STAMP   // diagnostic builds (scripts/experiments/stream_stamps.py): wall-clock stamps of every workgroup's phases;
                    // -DSTREAM_STAMP_SEL = 0 propup (default) | 1 propdown | 2 statistics GEMM
#ifndef STREAM_STAMP_SEL
#define STREAM_STAMP_SEL 0
#endif
#define ST_STAMP(SLOT)                                                                                                      \
    do {                                                                                                                    \
        if ((STREAM_STAMP_SEL == 0 ? (LA == LAY_K && LB == LAY_MN) : STREAM_STAMP_SEL == 1 ? (LA == LAY_K && LB == LAY_K)   \
                                                                                           : (LA == LAY_MN && LB == LAY_MN)) && \
            g.stamps && threadIdx.x == 0)                                                                                   \
            g.stamps[(int64_t)blockIdx.x * 8 + (SLOT)] = wall_clock64();                                                    \
    } while (0)
#else
#define ST_STAMP(SLOT) do {} while (0)
#endif

// AP = 3: six piece products; AP = 1: the row operand holds 0/1 values (one bf16 piece: the upper halves), three products
template <int LA, int LB, int MI, int NI, int FUSED, int AP>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void stream_gemm_kernel(GemmArgs g)
{
    MDBN_GEMM_ARGS_EARLY(g);
    ST_STAMP(0);
    constexpr int NW = SKINNY_WAVES, BM = 32 * MI;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [NW][BM][SKINNY_LDT] (+ 8)
    // workgroup -> tile, XCD-aware: consecutive workgroup ids go round the 8 XCDs, each with its own L2; XCD x takes a
    // CONTIGUOUS range of the row-major tile list (a few row tiles x all column strips), so its L2 holds those rows of the
    // row operand + the column operand once, instead of every XCD streaming both operands whole (7 MB through 4 MB of L2)
    const int nb = (int)gridDim.x;
    const int xcd = (int)blockIdx.x & 7, xq = (int)blockIdx.x >> 3;
    const int bid = xcd * (nb >> 3) + min(xcd, nb & 7) + xq;
    const int tn = (g.tiles_n + NI - 1) / NI;             // column tiles (g.tiles_n counts 32-column strips)
    const int per_split = tn * g.tiles_m;
    const int ks = bid / per_split, rem = bid - ks * per_split;
    const int tm = rem / tn, st = rem - tm * tn;
    const int m0 = tm * BM, n0 = st * 32 * NI;
    const int kbeg = ks * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;

    const float* aptr[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        const int am = min(m0 + i + 32 * a, g.M - 1);
        aptr[a] = LA == LAY_K ? g.A + (int64_t)am * g.lda : g.A + am;
    }
    const float* bptr[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int bn = min(n0 + i + 32 * b, g.N - 1);
        bptr[b] = LB == LAY_K ? g.B + (int64_t)bn * g.ldb : g.B + bn;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    if constexpr (LA == LAY_MN && LB == LAY_MN) {
        // statistics GEMM: the bias statistics / cost total / bias update units spread over the waves of all workgroups,
        // ahead of the operand stream (as skinny_gemm_kernel; behind the first loads they cost 60 registers and 6 - 13 us
        // per step: profiles/r05y_stream_variants.log)
        if (g.fin_enabled) {
            const int nu = fin_units(g.fin);
            for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += NW * (int)gridDim.x)
                finalize_unit(g.fin, unit, lane);
        }
    }
    ST_STAMP(1);
    float fa0[MI][8], fa1[MI][8], fa2[MI][8], fb0[NI][8], fb1[NI][8], fb2[NI][8];
    const int nfull = (kend - kbeg) >> 4;                              // whole 16-deep steps; wave w: steps w, w + 8, ...
    const int n = wave < nfull ? (nfull - wave + NW - 1) / NW : 0;     // (wave-uniform)
    auto issue = [&](int j, float (&fa)[MI][8], float (&fb)[NI][8]) {
        const int k16 = kbeg + 16 * (wave + NW * j) + 8 * h;
#pragma unroll
        for (int a = 0; a < MI; ++a) stream_load8<LA>(aptr[a], g.lda, k16, fa[a]);
#pragma unroll
        for (int b = 0; b < NI; ++b) stream_load8<LB>(bptr[b], g.ldb, k16, fb[b]);
    };
    auto consume = [&](const float (&fa)[MI][8], const float (&fb)[NI][8]) {
        tbf16x8 pb[NI][3];
#pragma unroll
        for (int b = 0; b < NI; ++b) th_split8(fb[b], pb[b]);
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            tbf16x8 pa[3];
            if constexpr (AP == 1) {
                tu32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    q[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, fa[a][2 * e + 1]), __builtin_bit_cast(unsigned, fa[a][2 * e]),
                                                 0x07060302u);
                pa[0] = __builtin_bit_cast(tbf16x8, q); pa[1] = pa[0]; pa[2] = pa[0];
            } else {
                th_split8(fa[a], pa);
            }
#pragma unroll
            for (int b = 0; b < NI; ++b) th_mma<AP>(acc[a][b], pa, pb[b]);
        }
    };
    int j = 0;
    if (n >= 6) {
        // steady state: straight-line code, two steps of loads in flight behind the one being multiplied
        issue(0, fa0, fb0); issue(1, fa1, fb1); issue(2, fa2, fb2);
        for (; j + 6 <= n; j += 3) {        // (the scheduling fences keep hipcc from sinking all three issues to the loop's end)
            consume(fa0, fb0); __builtin_amdgcn_sched_barrier(0); issue(j + 3, fa0, fb0); __builtin_amdgcn_sched_barrier(0);
            consume(fa1, fb1); __builtin_amdgcn_sched_barrier(0); issue(j + 4, fa1, fb1); __builtin_amdgcn_sched_barrier(0);
            consume(fa2, fb2); __builtin_amdgcn_sched_barrier(0); issue(j + 5, fa2, fb2); __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        if (n > 0) issue(0, fa0, fb0);
        if (n > 1) issue(1, fa1, fb1);
        if (n > 2) issue(2, fa2, fb2);
    }
    ST_STAMP(2);
    // drain (and short ranges): the sets hold steps j, j + 1, j + 2
    for (; j < n; j += 3) {
        consume(fa0, fb0);
        if (j + 3 < n) issue(j + 3, fa0, fb0);
        if (j + 1 < n) consume(fa1, fb1);
        if (j + 4 < n) issue(j + 4, fa1, fb1);
        if (j + 2 < n) consume(fa2, fb2);
        if (j + 5 < n) issue(j + 5, fa2, fb2);
    }
    if (((kend - kbeg) & 15) && wave == nfull % NW) {                  // the partial last step: one wave
        const int k16 = kbeg + 16 * nfull + 8 * h;
#pragma unroll
        for (int a = 0; a < MI; ++a) stream_load8_tail<LA>(aptr[a], g.lda, k16, kend, fa0[a]);
#pragma unroll
        for (int b = 0; b < NI; ++b) stream_load8_tail<LB>(bptr[b], g.ldb, k16, kend, fb0[b]);
        consume(fa0, fb0);
    }
    ST_STAMP(3);
    // the epilogue runs strip by strip (the partial tiles of one 32-column strip are parked, reduced, consumed)
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        if (b > 0) {
            if (n0 + 32 * b >= g.Nst) break;                       // (block-uniform: the last column tile may hold one strip)
            __syncthreads();                                        // every thread has read the previous strip's partials
        }
        f32x16 t[MI];
#pragma unroll
        for (int a = 0; a < MI; ++a) t[a] = acc[a][b];
        skinny_tile_epilogue<MI, FUSED, true>(g, t, smem, ks, m0, n0 + 32 * b, (ks * g.tiles_m + tm) * g.tiles_n + NI * st + b);
        ST_STAMP(4 + b);
    }
}

template <int LA, int LB, int MI, int NI, int FUSED, int AP>
static hipError_t launch_stream_t(const GemmArgs& g, hipStream_t s)
{
    constexpr int lds_bytes = (SKINNY_WAVES * 32 * MI * SKINNY_LDT + 8) * (int)sizeof(float);
    static bool attr_set = false;
    auto kern = stream_gemm_kernel<LA, LB, MI, NI, FUSED, AP>;
    if (!attr_set && lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(((g.tiles_n + NI - 1) / NI) * g.tiles_m * g.splitk), dim3(64 * SKINNY_WAVES), lds_bytes, s, g);
    return hipGetLastError();
}

// g.skinny = 1 with g.x6 = 1 (six products) | 2 (row operand 0/1: three)
hipError_t launch_stream_gemm(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (g.M < 1 || (g.mi != 1 && g.mi != 2) || (g.ni != 1 && g.ni != 2) || (int64_t)g.tiles_m * 32 * g.mi < g.M || g.kchunk % 16 != 0 ||
        (g.fused && g.splitk != 1) || (g.x6 != 1 && g.x6 != 2) || (int64_t)g.tiles_n * 32 < g.Nst)
        return hipErrorInvalidValue;
    const int ap = g.x6 == 2 ? 1 : 3;
#define STREAM_CASE(LAV, LBV, MIV, NIV, FV, APV) \
    if (la == LAV && lb == LBV && g.mi == MIV && g.ni == NIV && g.fused == FV && ap == APV) return launch_stream_t<LAV, LBV, MIV, NIV, FV, APV>(g, s)
#define STREAM_FWD(LBV, MIV, NIV) \
    STREAM_CASE(LAY_K, LBV, MIV, NIV, 0, 3); STREAM_CASE(LAY_K, LBV, MIV, NIV, 1, 3); STREAM_CASE(LAY_K, LBV, MIV, NIV, 0, 1); STREAM_CASE(LAY_K, LBV, MIV, NIV, 1, 1)
    // forward passes: plain / activation epilogue, general / 0-1 row operand
    STREAM_FWD(LAY_K, 1, 1);  STREAM_FWD(LAY_K, 2, 1);  STREAM_FWD(LAY_K, 2, 2);
    STREAM_FWD(LAY_MN, 1, 1); STREAM_FWD(LAY_MN, 2, 1); STREAM_FWD(LAY_MN, 2, 2);
    // statistics GEMM: plain / parameter update
    STREAM_CASE(LAY_MN, LAY_MN, 1, 1, 0, 3); STREAM_CASE(LAY_MN, LAY_MN, 1, 1, 2, 3);
    STREAM_CASE(LAY_MN, LAY_MN, 2, 1, 0, 3); STREAM_CASE(LAY_MN, LAY_MN, 2, 1, 2, 3);
    STREAM_CASE(LAY_MN, LAY_MN, 2, 2, 0, 3); STREAM_CASE(LAY_MN, LAY_MN, 2, 2, 2, 3);
#undef STREAM_FWD
#undef STREAM_CASE
    return hipErrorInvalidValue;
}

}  // namespace mdbn
