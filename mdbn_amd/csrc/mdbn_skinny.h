// The tile epilogue of the register-streaming GEMMs (skinny_gemm_kernel in mdbn_kernels.hip, stream_gemm_kernel in
// mdbn_stream.hip): the 8 waves' partial accumulators of a (32 MI) x 32 output tile are parked in LDS, reduced in wave order
// (deterministic) and handed to the epilogue the launch asked for.
#pragma once
#include <hip/hip_runtime.h>
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_bf16x3.h"

namespace mdbn {

constexpr int SKINNY_WAVES = 8, SKINNY_LDT = 33;

// The kernel arguments a GEMM kernel needs before its first global access, requested in ONE batch at the top: hipcc loads
// kernarg fields where they are first used, which put two to four scalar-load round trips in a row at the head of every
// launch (plane GEMMs: same-box 143.3 -> 141.6 us per headline step, profiles/r04zy_args_early_ab.log).
#define MDBN_GEMM_ARGS_EARLY(G)                                                                                         \
    asm volatile("" :: "s"((G).A), "s"((G).B), "s"((G).C), "s"((G).lda), "s"((G).ldb), "s"((G).ldc), "s"((G).slab_stride),   \
                 "s"((G).M), "s"((G).N), "s"((G).K), "s"((G).Nst), "s"((G).kchunk), "s"((G).splitk), "s"((G).tiles_m),       \
                 "s"((G).tiles_n), "s"((int)gridDim.x))


// FUSED 0: split-K slab / plain C; 1: bias + activation + sampling (act_quad); 2: the parameter update (statistics GEMM,
// update_rule4).  smem: [SKINNY_WAVES][32 MI][SKINNY_LDT] floats (+ 8).  slot: where a FUSED 1 launch with a cost target
// leaves this tile's cost partial.
#ifdef MDBN_STAMP   // diagnostic builds: the epilogue's own phases, slots 8.. of the workgroup's 16 (stream_stamps.py)
#define SK_STAMP(SLOT) do { if (HOIST && g.stamps && threadIdx.x == 0 && n0 == sk_first_n0) g.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64(); } while (0)
#else
#define SK_STAMP(SLOT) do {} while (0)
#endif

template <int MI, int FUSED, bool HOIST>
__device__ __forceinline__ void skinny_tile_epilogue(const GemmArgs& g, const f32x16 (&acc)[MI], float* smem, int ks, int m0, int n0, int slot)
{
    constexpr int NW = SKINNY_WAVES, LDT = SKINNY_LDT, BM = 32 * MI;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int q = threadIdx.x;                 // quad = (row group, column) of the tile
    const int rg = q >> 5, c = q & 31;
    const int col = n0 + c;
    const int r0w = m0 + 4 * rg;

    // HOIST: the epilogue's own global operands are requested FIRST: their round trips pass under the parking of the partial
    // tiles and the barrier instead of behind them (streaming bf16x6 kernel: -1 .. -3 us per step; the exact-f32 kernel of
    // the small layers loses 3.6 us per CD-5 step at 256 -> 200 with it: profiles/r05y_stream_variants.log)
    float bias = 0.f, tg4[4] = {0.f, 0.f, 0.f, 0.f};
    float wv[4] = {}, sv[4] = {}, w0v[4] = {};
    bool on = false;
    auto request = [&]() {
    if constexpr (FUSED == 1) {
        const EpiArgs& e = g.epi;
        on = rg < BM / 4 && r0w < e.rows && col < (int)e.ld;
        if (on) {
            const bool live = col < e.cols;
            bias = live ? e.bias[col] : 0.f;
            act_quad_targets(e, r0w, col, live, tg4);
        }
    } else if constexpr (FUSED == 2) {
        const UpdEpi& u = g.upd;
        on = rg < BM / 4 && r0w < u.rows && col < (int)u.ld;
        if (on) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = r0w + j < u.rows;
                const int64_t off = (int64_t)(ok ? r0w + j : r0w) * u.ld + col;
                wv[j] = u.W[off];
                sv[j] = u.Ws[off];
                w0v[j] = u.W0 ? u.W0[off] : 0.f;
            }
        }
    }
    };
#ifdef MDBN_STAMP
    const int sk_first_n0 = g.stamps ? (int)g.stamps[(int64_t)blockIdx.x * 16 + 7] : -1;     // (set by the kernel: strip 0's n0)
#endif
    SK_STAMP(8);
    if (HOIST) request();

    // park the partial accumulators, reduce over the waves in wave order
    float* T = smem + wave * (BM * LDT);
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            T[(32 * a + (e & 3) + 8 * (e >> 2) + 4 * h) * LDT + i] = acc[a][e];
    SK_STAMP(9);
    __syncthreads();
    SK_STAMP(10);
    if (!HOIST) request();
    if (HOIST) {
        // (complete here -- the barrier waited for them -- but hipcc does not carry that across the per-row branches below
        // and would wait for ALL memory operations, the previous row's stores included, before every row: see
        // stream_tile_epilogue2)
        asm volatile("" : "+v"(bias));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("" : "+v"(tg4[j]));
            if constexpr (FUSED == 2) asm volatile("" : "+v"(wv[j]), "+v"(sv[j]), "+v"(w0v[j]));
        }
    }

    float cost = 0.f;
    if (rg < BM / 4) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += smem[w * (BM * LDT) + (4 * rg + j) * LDT + c];
            x[j] = sum;
        }
        SK_STAMP(11);
        if constexpr (FUSED == 1) {
            const EpiArgs& e = g.epi;
            if (on) {
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] += bias;
                act_quad_tg(e, x[0], x[1], x[2], x[3], r0w, col, col < e.cols, cost, tg4);
            }
        } else if constexpr (FUSED == 2) {
            const UpdEpi& u = g.upd;
            if (on) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!u.W0) w0v[j] = wv[j];
                    if (col >= g.N) x[j] = 0.f;            // pad columns: S is exactly zero there
                }
                float4 wn, sn;
                update_rule4(make_float4(wv[0], wv[1], wv[2], wv[3]), make_float4(sv[0], sv[1], sv[2], sv[3]),
                             make_float4(x[0], x[1], x[2], x[3]), make_float4(w0v[0], w0v[1], w0v[2], w0v[3]),
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
    }
    SK_STAMP(12);
    if constexpr (FUSED == 1) {
        if (g.epi.cost_partials) {
            __syncthreads();
            const float tot = block_sum(cost, smem);
            if (threadIdx.x == 0) g.epi.cost_partials[slot] = tot;          // slot: (K range, row tile, strip)
        }
    }
}

}  // namespace mdbn
