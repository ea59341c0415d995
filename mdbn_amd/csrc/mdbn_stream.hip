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
#include <algorithm>
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

// The tile epilogue of a 64-column tile (NI = 2) in ONE pass: both strips' partial tiles are parked together (one barrier
// instead of two park / barrier / reduce rounds: ~2 us per launch at the 2048 -> 400 propdown, profiles/r05zb), a thread
// owns the same (row group, column) quad in each strip.  Arithmetic per quad: skinny_tile_epilogue's.
// smem: [SKINNY_WAVES][2][32 MI][SKINNY_LDT] floats (+ 8).  One cost partial per tile, in strip 0's slot (strip 1's: zero).
#ifdef MDBN_STAMP
#define E2_STAMP(SLOT) do { if (g.stamps && threadIdx.x == 0) g.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64(); } while (0)
#else
#define E2_STAMP(SLOT) do {} while (0)
#endif
template <int MI, int FUSED>
__device__ __forceinline__ void stream_tile_epilogue2(const GemmArgs& g, const f32x16 (&acc)[MI][2], float* smem, int ks, int m0, int n0, int slot0)
{
    constexpr int NW = SKINNY_WAVES, LDT = SKINNY_LDT, BM = 32 * MI, ST = BM * LDT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int q = threadIdx.x;
    const int rg = q >> 5, c = q & 31;
    const int r0w = m0 + 4 * rg;
    const bool strip1 = n0 + 32 < g.Nst;                   // (block-uniform: the last column tile may hold one strip)

    E2_STAMP(8);
    float bias[2] = {0.f, 0.f}, tg4[2][4];
    float wv[2][4] = {}, sv[2][4] = {}, w0v[2][4] = {};
    bool on[2] = {false, false};
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = n0 + 32 * b + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) tg4[b][j] = 0.f;
        if constexpr (FUSED == 1) {
            const EpiArgs& e = g.epi;
            on[b] = rg < BM / 4 && r0w < e.rows && col < (int)e.ld;
            if (on[b]) {
                const bool live = col < e.cols;
                bias[b] = live ? e.bias[col] : 0.f;
                act_quad_targets(e, r0w, col, live, tg4[b]);
            }
        } else if constexpr (FUSED == 2) {
            const UpdEpi& u = g.upd;
            on[b] = rg < BM / 4 && r0w < u.rows && col < (int)u.ld;
            if (on[b]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = r0w + j < u.rows;
                    const int64_t off = (int64_t)(ok ? r0w + j : r0w) * u.ld + col;
                    wv[b][j] = u.W[off];
                    sv[b][j] = u.Ws[off];
                    w0v[b][j] = u.W0 ? u.W0[off] : 0.f;
                }
            }
        }
    }

    float* T = smem + wave * (2 * ST);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                T[b * ST + (32 * a + (e & 3) + 8 * (e >> 2) + 4 * h) * LDT + i] = acc[a][b][e];
    E2_STAMP(9);
    __syncthreads();
    E2_STAMP(10);
    // The requested operands are complete here (the barrier waited for them), but hipcc does not carry that fact across
    // the per-row branches of the epilogue: it put s_waitcnt vmcnt(0) in front of every row's use of them -- which also
    // waits for the PREVIOUS row's stores to be acknowledged, ~0.5 us per row (4.6 of the 11.3 us of the 2048 -> 400
    // propdown).  Re-defining the registers through an empty asm cuts them loose from their loads.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        asm volatile("" : "+v"(bias[b]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("" : "+v"(tg4[b][j]));
            if constexpr (FUSED == 2) asm volatile("" : "+v"(wv[b][j]), "+v"(sv[b][j]), "+v"(w0v[b][j]));
        }
    }

    float cost = 0.f;
    if (rg < BM / 4) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b == 1 && !strip1) break;
            const int col = n0 + 32 * b + c;
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += smem[w * (2 * ST) + b * ST + (4 * rg + j) * LDT + c];
                x[j] = sum;
            }
            E2_STAMP(b == 0 ? 11 : 14);
            if constexpr (FUSED == 1) {
                const EpiArgs& e = g.epi;
                if (on[b]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] += bias[b];
                    act_quad_tg(e, x[0], x[1], x[2], x[3], r0w, col, col < e.cols, cost, tg4[b]);
                }
            } else if constexpr (FUSED == 2) {
                const UpdEpi& u = g.upd;
                if (on[b]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!u.W0) w0v[b][j] = wv[b][j];
                        if (col >= g.N) x[j] = 0.f;            // pad columns: S is exactly zero there
                    }
                    float4 wn, sn;
                    update_rule4(make_float4(wv[b][0], wv[b][1], wv[b][2], wv[b][3]), make_float4(sv[b][0], sv[b][1], sv[b][2], sv[b][3]),
                                 make_float4(x[0], x[1], x[2], x[3]), make_float4(w0v[b][0], w0v[b][1], w0v[b][2], w0v[b][3]),
                                 u.inv_bs, u.wc, upd_decay(u.lr, u.l2), u.l1, upd_two_lr_l1(u.lr, u.l1), u.mu, u.lr, wn, sn);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (r0w + j < u.rows) {
                            const int64_t off = (int64_t)(r0w + j) * u.ld + col;
                            u.W[off] = comp(wn, j);
                            u.Ws[off] = comp(sn, j);
                            if (u.Wp) {
                                unsigned short p1, p2, p3;
                                split3(comp(wn, j), p1, p2, p3);
                                u.Wp[off] = p1; u.Wp[u.wp_stride + off] = p2; u.Wp[2 * u.wp_stride + off] = p3;
                            }
                        }
                }
            } else {
                float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0w + j < g.M && col < g.Nst) C[(int64_t)(r0w + j) * g.ldc + col] = col < g.N ? x[j] : 0.f;
            }
            if (b == 0) E2_STAMP(13);
        }
    }
    E2_STAMP(12);
    if constexpr (FUSED == 1) {
        if (g.epi.cost_partials) {
            __syncthreads();
            const float tot = block_sum(cost, smem);
            if (threadIdx.x == 0) {
                g.epi.cost_partials[slot0] = tot;
                if (strip1) g.epi.cost_partials[slot0 + 1] = 0.f;
            }
        }
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
            g.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64();                                                    \
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
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [NW][NI][BM][SKINNY_LDT] (+ 8)
    const int tn = (g.tiles_n + NI - 1) / NI;             // column tiles (g.tiles_n counts 32-column strips)
    const int per_split = tn * g.tiles_m;
    const int nb = per_split * g.splitk;                  // tile workgroups
    if constexpr (LA == LAY_MN && LB == LAY_MN) {
        // statistics GEMM: the bias statistics / cost total / bias update units (finalize_unit: the functions and order of
        // finalize_stats_kernel) run on EXTRA workgroups behind the tile workgroups -- ahead of the operand stream inside
        // the tile workgroups they held some of them up by 3 - 6 us (profiles/r05w_stream_stamps.log)
        if ((int)blockIdx.x >= nb) {
            if (g.fin_enabled) {
                const int nu = fin_units(g.fin), nx = (int)gridDim.x - nb, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                for (int unit = wv * nx + ((int)blockIdx.x - nb); unit <= nu; unit += SKINNY_WAVES * nx)
                    finalize_unit(g.fin, unit, threadIdx.x & 63);
            }
            return;
        }
    }
    // workgroup -> tile, XCD-aware: consecutive workgroup ids go round the 8 XCDs, each with its own L2; XCD x takes a
    // CONTIGUOUS range of the row-major tile list (a few row tiles x all column strips), so its L2 holds those rows of the
    // row operand + the column operand once, instead of every XCD streaming both operands whole (7 MB through 4 MB of L2)
    const int xcd = (int)blockIdx.x & 7, xq = (int)blockIdx.x >> 3;
    const int bid = xcd * (nb >> 3) + min(xcd, nb & 7) + xq;
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

    ST_STAMP(1);
    // D register sets: D - 1 steps of loads in flight behind the one being multiplied.  Deeper does not help (D = 4 / 5 / 6
    // on the 32 x 32 tiles: +0 .. +4 % per step, profiles/r05zo_stream_depth.log): the stream moves ~11 TB/s from L2 into
    // the CUs, it is not short of requests in flight.
#ifndef STREAM_DEPTH
#define STREAM_DEPTH 3
#endif
    constexpr int D = MI * NI == 1 ? STREAM_DEPTH : 3;
    float fa[D][MI][8], fb[D][NI][8];               // (indexed by unrolled loop counters only)
    const int nfull = (kend - kbeg) >> 4;                              // whole 16-deep steps; wave w: steps w, w + 8, ...
    const int n = wave < nfull ? (nfull - wave + NW - 1) / NW : 0;     // (wave-uniform)
    auto issue = [&](int j, float (&xa)[MI][8], float (&xb)[NI][8]) {
        const int k16 = kbeg + 16 * (wave + NW * j) + 8 * h;
#pragma unroll
        for (int a = 0; a < MI; ++a) stream_load8<LA>(aptr[a], g.lda, k16, xa[a]);
#pragma unroll
        for (int b = 0; b < NI; ++b) stream_load8<LB>(bptr[b], g.ldb, k16, xb[b]);
    };
    auto consume = [&](const float (&xa)[MI][8], const float (&xb)[NI][8]) {
        tbf16x8 pb[NI][3];
#pragma unroll
        for (int b = 0; b < NI; ++b) th_split8(xb[b], pb[b]);
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            tbf16x8 pa[3];
            if constexpr (AP == 1) {
                tu32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    q[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, xa[a][2 * e + 1]), __builtin_bit_cast(unsigned, xa[a][2 * e]),
                                                 0x07060302u);
                pa[0] = __builtin_bit_cast(tbf16x8, q); pa[1] = pa[0]; pa[2] = pa[0];
            } else {
                th_split8(xa[a], pa);
            }
#pragma unroll
            for (int b = 0; b < NI; ++b) th_mma<AP>(acc[a][b], pa, pb[b]);
        }
    };
    int j = 0;
    if (n >= 2 * D) {
        // steady state: straight-line code (loads under data-dependent branches would cost the wait-count precision)
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, fa[d], fb[d]);
        for (; j + 2 * D <= n; j += D) {     // (the scheduling fences keep hipcc from sinking all the issues to the loop's end)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                consume(fa[d], fb[d]); __builtin_amdgcn_sched_barrier(0);
                issue(j + D + d, fa[d], fb[d]); __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (n > d) issue(d, fa[d], fb[d]);
    }
    ST_STAMP(2);
    // drain (and short ranges): the sets hold steps j .. j + D - 1
    for (; j < n; j += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (j + d < n) consume(fa[d], fb[d]);
            if (j + D + d < n) issue(j + D + d, fa[d], fb[d]);
        }
    }
    if (((kend - kbeg) & 15) && wave == nfull % NW) {                  // the partial last step: one wave
        const int k16 = kbeg + 16 * nfull + 8 * h;
#pragma unroll
        for (int a = 0; a < MI; ++a) stream_load8_tail<LA>(aptr[a], g.lda, k16, kend, fa[0][a]);
#pragma unroll
        for (int b = 0; b < NI; ++b) stream_load8_tail<LB>(bptr[b], g.ldb, k16, kend, fb[0][b]);
        consume(fa[0], fb[0]);
    }
    ST_STAMP(3);
#ifdef MDBN_STAMP
    if (g.stamps && threadIdx.x == 0) g.stamps[(int64_t)blockIdx.x * 16 + 7] = (unsigned long long)n0;
    __syncthreads();
#endif
    if constexpr (NI == 2) {
        stream_tile_epilogue2<MI, FUSED>(g, acc, smem, ks, m0, n0, (ks * g.tiles_m + tm) * g.tiles_n + NI * st);
        ST_STAMP(4);
    } else {
        f32x16 t[MI];
#pragma unroll
        for (int a = 0; a < MI; ++a) t[a] = acc[a][0];
        skinny_tile_epilogue<MI, FUSED, true>(g, t, smem, ks, m0, n0, (ks * g.tiles_m + tm) * g.tiles_n + st);
        ST_STAMP(4);
    }
}

template <int LA, int LB, int MI, int NI, int FUSED, int AP>
static hipError_t launch_stream_t(const GemmArgs& g, hipStream_t s)
{
    constexpr int lds_bytes = (SKINNY_WAVES * NI * 32 * MI * SKINNY_LDT + 8) * (int)sizeof(float);
    static bool attr_set = false;
    auto kern = stream_gemm_kernel<LA, LB, MI, NI, FUSED, AP>;
    if (!attr_set && lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int blocks = ((g.tiles_n + NI - 1) / NI) * g.tiles_m * g.splitk;
    if (LA == LAY_MN && LB == LAY_MN && g.fin_enabled) {
        const int units = (int)((g.fin.ldh + g.fin.ldv + 15) / 16) + 1;                   // fin_units() + the cost unit
        blocks += std::min(32, (units + SKINNY_WAVES - 1) / SKINNY_WAVES);                // one unit per wave, <= 32 workgroups
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SKINNY_WAVES), lds_bytes, s, g);
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
