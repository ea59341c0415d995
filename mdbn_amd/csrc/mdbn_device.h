// Device-side helpers shared by the GEMM kernels of mdbn_kernels.hip and mdbn_planes.hip: the update rule
// (rbm.py:347-365), the finalize units (bias statistics / cost / bias update), the activation + sampling
// epilogue on a tile parked in LDS, and the fused parameter update of the statistics GEMM.  Header-only so
// that each translation unit keeps its own register allocation (cdna_hip_programming.md rule 19).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"
#include "mdbn_kernels.h"

namespace mdbn {

// ----------------------------------------------------------------------------------
// float4 component access and the parameter-update rule (used by the GEMM's fused epilogues too)
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float comp(const float4& v, int j)
{
    return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}
__device__ __forceinline__ void setc(float4& v, int j, float x)
{
    if (j == 0) v.x = x; else if (j == 1) v.y = x; else if (j == 2) v.z = x; else v.w = x;
}

// The update rule (rbm.py:347-365) on 4 weights, shared by update_kernel and the fused epilogues of
// the statistics GEMM.  Floating-point contraction is switched off inside these helpers (hipcc
// otherwise fuses a*b+c into an fma or not depending on what the helper is inlined into -- the
// __f*_rn intrinsics are plain operators to it), so every call site agrees bit for bit, and with the
// float32 restatement, which rounds every operation.
__device__ __forceinline__ float upd_grad(float st, float inv_bs, float wc, float w0)
{
#pragma clang fp contract(off)
    const float a = st * inv_bs, b = wc * w0;
    return a - b;
}
__device__ __forceinline__ float upd_speed(float g, float sp, float mu)     // g + (s - g) * mu
{
#pragma clang fp contract(off)
    const float d = sp - g;
    const float e = d * mu;
    return g + e;
}
__device__ __forceinline__ float upd_param(float w, float m, float sp, float lr)   // w * m + s_old * lr
{
#pragma clang fp contract(off)
    const float a = w * m, b = sp * lr;
    return a + b;
}
__device__ __forceinline__ float upd_scale(float x, float s)
{
#pragma clang fp contract(off)
    return x * s;
}
__device__ __forceinline__ float upd_decay(float lr, float l2)          // 1 - 2 lr l2
{
#pragma clang fp contract(off)
    const float a = 2.0f * lr;
    const float b = a * l2;
    return 1.0f - b;
}
__device__ __forceinline__ float upd_two_lr_l1(float lr, float l1)
{
#pragma clang fp contract(off)
    const float a = 2.0f * lr;
    return a * l1;
}
__device__ __forceinline__ float upd_shrink(float two_lr_l1, float w)    // 1 + 2 lr l1 / (|w| + eps)
{
#pragma clang fp contract(off)
    const float d = fabsf(w) + 0.001f;
    const float q = __fdiv_rn(two_lr_l1, d);
    return 1.0f + q;
}

__device__ __forceinline__ void update_rule4(const float4& w, const float4& sp, const float4& st, const float4& wc0,
                                             float inv_bs, float wc, float decay, float l1, float two_lr_l1,
                                             float mu, float lr, float4& wn, float4& sn)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float wj = comp(w, j), spj = comp(sp, j);
        float g = upd_grad(comp(st, j), inv_bs, wc, comp(wc0, j));
        float m = decay;
        if (l1 != 0.0f) {
            const float shrink = upd_shrink(two_lr_l1, wj);
            g = __fdiv_rn(g, shrink);
            m = __fdiv_rn(decay, shrink);
        }
        setc(sn, j, upd_speed(g, spj, mu));
        setc(wn, j, upd_param(wj, m, spj, lr));
    }
}

// ----------------------------------------------------------------------------------
// Bias statistics (rbm.py:416-417) from the epilogues' 4-row column partials, the cost total, and the
// bias half of the update, in units one WAVE computes on its own (no LDS, no barrier): unit u < n_units
// = 16 columns x 4 group quarters (lane = 4 * column + quarter; quarter sums run over ascending groups,
// combined as (q0 + q1) + (q2 + q3)); unit n_units = the cost total.  Fixed order: deterministic, and
// identical whether finalize_stats_kernel or the statistics GEMM's consumer waves run the units.
// ----------------------------------------------------------------------------------
__device__ __forceinline__ int fin_units(const FinArgs& f) { return (int)((f.ldh + f.ldv + 15) / 16); }

// column i of [s_h | s_v] is known: store it and apply the bias half of the update (rbm.py:356-365; same helpers as update_kernel)
__device__ __forceinline__ void finalize_column(const FinArgs& f, int64_t i, float t)
{
    if (i < f.ldh) f.s_h[i] = t;
    else if (i < f.ldh + f.ldv) f.s_v[i - f.ldh] = t;
    if (f.do_bias) {
        const BiasUpd& bu = f.bu;
        // (speed and parameter loaded TOGETHER: read one after the other -- load, wait, store, load, wait, store -- they were
        //  two memory round trips at the tail of every unit)
        if (i < bu.H) {
            const float sp = bu.hbs[i], p0 = bu.hb[i];
            bu.hbs[i] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
            bu.hb[i] = upd_param(p0, 1.0f, sp, bu.lr);
        } else if (i >= f.ldh && i - f.ldh < bu.V) {
            const int64_t j = i - f.ldh;
            const float sp = bu.vbs[j], p0 = bu.vb[j];
            bu.vbs[j] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
            bu.vb[j] = upd_param(p0, 1.0f, sp, bu.lr);
        }
    }
}

__device__ __forceinline__ void finalize_unit(const FinArgs& f, int unit, int lane)
{
    const int n_units = fin_units(f);
    if (unit < n_units) {
        const int cl = lane >> 2, qd = lane & 3;
        const int64_t i = (int64_t)unit * 16 + cl;
        const int per = (f.ngroups + 3) / 4;
        const int gbeg = qd * per, gend = min(f.ngroups, gbeg + per);
        float a = 0.f;
        // (hipcc predicates every load of the plain loop into its own block and waits for them in pairs -- `s_waitcnt
        //  vmcnt(2) / (0)` between 2-load groups in the ISA; a 32-wide unroll had made the unit take 11 us instead of 3.4,
        //  round 3.  The batched form -- loads of 8 groups issued together at clamped addresses, dead entries added as an
        //  exact zero, the same sums -- is the A/B alternative.)
#ifndef MDBN_FIN_BATCH
#define MDBN_FIN_BATCH 0   // 1: batches of 8 groups with clamped addresses (below).  Same-box A/B, profiles/r04zu_fin_batch_ab.log:
                           // c1 (784 -> 500, batch 20: 5 groups) 41.5 -> 40.7 us per step, but the headline step 141.0 -> 142.1
                           // (128 groups: 16 more loads per lane in flight beside the first LDS-DMA stages of the statistics
                           // GEMM, whose MFMA waves run these units) -- the plain loop stays
#endif
#if !MDBN_FIN_BATCH
        if (i < f.ldh) {
#pragma unroll 16
            for (int g = gbeg; g < gend; ++g) a += f.posP[(int64_t)g * f.ldh + i] + f.negP[(int64_t)g * f.ldh + i];
        } else if (i < f.ldh + f.ldv) {
            const int64_t j = i - f.ldh;
#pragma unroll 16
            for (int g = gbeg; g < gend; ++g) a += f.partV[(int64_t)g * f.ldv + j];
        }
#else
        if (i < f.ldh) {
            for (int g0 = gbeg; g0 < gend; g0 += 8) {
                float p[8], q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t g = min(g0 + u, gend - 1);
                    p[u] = f.posP[g * f.ldh + i]; q[u] = f.negP[g * f.ldh + i];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) a += g0 + u < gend ? p[u] + q[u] : 0.f;
            }
        } else if (i < f.ldh + f.ldv) {
            const int64_t j = i - f.ldh;
            for (int g0 = gbeg; g0 < gend; g0 += 8) {
                float p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = f.partV[(int64_t)min(g0 + u, gend - 1) * f.ldv + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += g0 + u < gend ? p[u] : 0.f;
            }
        }
#endif
        const int l0 = lane & ~3;
        const float r0 = __shfl(a, l0, 64), r1 = __shfl(a, l0 + 1, 64), r2 = __shfl(a, l0 + 2, 64), r3 = __shfl(a, l0 + 3, 64);
        const float t = (r0 + r1) + (r2 + r3);
        if (qd == 0) finalize_column(f, i, t);
    } else if (unit == n_units && f.cost_partials) {
        float a = 0.f;
        for (int k0 = lane; k0 < f.n_cost; k0 += 64 * 8) {     // (same order of additions per lane; 8 loads in flight instead of one)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k0 + 64 * u < f.n_cost ? f.cost_partials[k0 + 64 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) {
            f.cost[0] = a; f.cost[1] = 0.f; f.cost[2] = 0.f; f.cost[3] = 0.f;
            if (f.do_bias && f.bu.cost_out) f.bu.cost_out[0] = a * f.bu.cost_scale;
        }
    }
}


// ----------------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------------
// sigmoid / softplus on the hardware transcendentals (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp
// each).  |error| of sigmoid <= ~1e-7 absolute: the exponent's argument rounding |x|*6e-8 is
// multiplied by sigmoid' = p(1-p) <= 1/4.  The libm-grade expf/division these replace made
// the epilogue VALU-bound (14 us per call at B*H = 512K elements).
// row of EpiArgs.target that output row `row` is compared with (identity, or through the minibatch index)
__device__ __forceinline__ int64_t epi_target_row(const EpiArgs& e, int64_t row)
{
    if (!e.target_idx) return row;
    int64_t s = e.target_idx64 ? reinterpret_cast<const int64_t*>(e.target_idx)[row]
                               : (int64_t)reinterpret_cast<const int32_t*>(e.target_idx)[row];
    if (s < 0) s += e.target_rows;
    return s < 0 ? 0 : (s >= e.target_rows ? e.target_rows - 1 : s);
}

// ... of the four rows r0 .. r0 + 3 (rows past the end: the last one).  The four index loads are issued TOGETHER, the
// arithmetic on them afterwards: four calls of epi_target_row compile to four blocks (one per index width) with a
// `s_waitcnt vmcnt(0)` each -- four memory round trips in a row at the head of every epilogue with a cost target (the
// "operands arrived after 2.0 us" of the propdown launch's phase stamps).
__device__ __forceinline__ void epi_target_rows4(const EpiArgs& e, int r0, int64_t (&srow)[4])
{
    int64_t raw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = r0 + j < e.rows ? r0 + j : e.rows - 1;
    if (!e.target_idx) {
#pragma unroll
        for (int j = 0; j < 4; ++j) srow[j] = raw[j];
        return;
    }
    if (e.target_idx64) {
        const int64_t* ix = reinterpret_cast<const int64_t*>(e.target_idx);
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = ix[raw[j]];
    } else {
        const int32_t* ix = reinterpret_cast<const int32_t*>(e.target_idx);
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = (int64_t)ix[raw[j]];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t t = raw[j];
        if (t < 0) t += e.target_rows;
        srow[j] = t < 0 ? 0 : (t >= e.target_rows ? e.target_rows - 1 : t);
    }
}

__device__ __forceinline__ float sigmoidf_(float x)
{
#ifndef MDBN_SIGMOID_RCP
#define MDBN_SIGMOID_RCP 1      // 1: v_rcp_f32 (1 ulp); 0: __frcp_rn, which HIP expands to the 12-instruction correctly-rounded division
#endif
#if MDBN_SIGMOID_RCP
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#else
    return __frcp_rn(1.0f + __expf(-x));
#endif
}
__device__ __forceinline__ float softplusf_(float x)
{
    // max(x, 0) + log1p(exp(-|x|)): the log argument is in (1, 2], no cancellation
    return fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x)));
}

__device__ __forceinline__ float block_sum(float v, float* red /* >= 4 floats */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int k = 0; k < nw; ++k) t += red[k];      // fixed order: deterministic
    return t;
}

// 4 uniform words for rows g0..g0+3 of column col (one Philox block when g0 % 4 == 0)
__device__ __forceinline__ void philox_rows4(const PhiloxKey& k, uint32_t draw, uint64_t g0,
                                             uint32_t col, uint32_t (&w)[4])
{
    uint32_t lo[4];
    philox4x32_10(col, (uint32_t)(g0 >> 2), draw, k.step, k.k0, k.k1, lo);
    const uint32_t ph = (uint32_t)(g0 & 3);
    if (ph == 0) {
        w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = lo[3];
    } else {
        uint32_t hi[4];
        philox4x32_10(col, (uint32_t)(g0 >> 2) + 1u, draw, k.step, k.k0, k.k1, hi);
        // rows g0..g0+3 straddle two blocks: element r is word (ph + r) of the 8 words lo|hi.
        // Selected with compile-time indices only (a runtime index would put lo/hi in scratch).
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t cand = j < 4 ? lo[j & 3] : hi[j & 3];
                v = (ph + (uint32_t)r == (uint32_t)j) ? cand : v;
            }
            w[r] = v;
        }
    }
}

// ----------------------------------------------------------------------------------
// Activation epilogue fused into the GEMM (jobs that need no split-K): the arithmetic of
// act_epilogue_kernel below applied to the block's own 128 x BN tile, which the consumer waves
// parked in LDS (row stride BN + 8); all 8 waves take part.  A thread owns one column and walks
// 4-row groups -- one Philox4x32-10 block per (group, column), as everywhere.  One cost partial
// per block (cost_partials[blockIdx.x]); column partials [row_group][col] as below.
// ----------------------------------------------------------------------------------
// One (4-row group, column) of an activation epilogue: x[j] = pre-activation (bias included) of
// row r0 + j.  Stores pre / mean / sample, the group's column partial, and adds to `cost`.
// exact three-way split of an f32 into bf16 pieces by truncation: x = p1 + p2 + p3 (bf16 keeps f32's exponent
// range, three 8-bit significands cover the 24 bits).  The upper half of an f32 IS a bf16.
__device__ __forceinline__ void split3(float a, unsigned short& p1, unsigned short& p2, unsigned short& p3)
{
    const unsigned ua = __builtin_bit_cast(unsigned, a);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra);
    const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u);
    p1 = (unsigned short)(ua >> 16); p2 = (unsigned short)(va >> 16);
    p3 = (unsigned short)(__builtin_bit_cast(unsigned, sa) >> 16);
}

// four consecutive values -> one 8-byte store per plane
__device__ __forceinline__ void store_planes4(unsigned short* P, int64_t plane_stride, int64_t off, const float4& v)
{
    unsigned short q[3][4];
    split3(v.x, q[0][0], q[1][0], q[2][0]);
    split3(v.y, q[0][1], q[1][1], q[2][1]);
    split3(v.z, q[0][2], q[1][2], q[2][2]);
    split3(v.w, q[0][3], q[1][3], q[2][3]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = q[p][0] | ((unsigned)q[p][1] << 16);
        w.y = q[p][2] | ((unsigned)q[p][3] << 16);
        *reinterpret_cast<uint2*>(P + p * plane_stride + off) = w;
    }
}

// MODE >= 0: e.gauss (bit 1) and "a sample is wanted" (bit 0) as compile-time facts; act_quad_dispatch branches once.
// the cost targets of a quad's four rows (index -> row: two dependent loads each), for a caller that wants them in flight
// before the pre-activations exist
__device__ __forceinline__ void act_quad_targets(const EpiArgs& e, int r0, int col, bool live, float (&tg4)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) tg4[j] = 0.f;
    if (e.target && live) {
        int64_t srow[4];
        epi_target_rows4(e, r0, srow);
#pragma unroll
        for (int j = 0; j < 4; ++j) tg4[j] = e.target[srow[j] * e.ld_target + col];
    }
}

template <int MODE = -1>
__device__ __forceinline__ void act_quad_tg(const EpiArgs& e, float x0, float x1, float x2, float x3, int r0, int col, bool live, float& cost,
                                            const float (&tg4)[4])
{
    const bool is_gauss = MODE < 0 ? e.gauss != 0 : (MODE & 2) != 0;
    const bool need_u = MODE < 0 ? (e.sample != nullptr || e.sample_plane != nullptr) : (MODE & 1) != 0;
    const bool need_z = need_u && is_gauss;
    uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
    if (need_u) {
        const uint64_t g0 = e.rng.row_offset + (uint64_t)r0;
        philox_rows4(e.rng, e.rng.draw, g0, (uint32_t)col, wa);
        if (need_z) philox_rows4(e.rng, e.rng.draw | MDBN_NORMAL_BIT, g0, (uint32_t)col, wb);
    }
    float csum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + j;
        float xj = j == 0 ? x0 : (j == 1 ? x1 : (j == 2 ? x2 : x3));
        if (row < e.rows) {
            const int64_t off = (int64_t)row * e.ld + col;
            float m, sv = 0.f;
            if (is_gauss) {
                m = xj;
                if (need_u) {
                    const float u1 = philox_u01(wa[j]), u2 = philox_u01(wb[j]);
                    sv = m + sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                }
            } else {
                m = sigmoidf_(xj);
                if (need_u) sv = philox_u01(wa[j]) < m ? 1.0f : 0.0f;
            }
            float tg = 0.f;
            if (e.target && live) {
                tg = tg4[j];
                if (is_gauss) { const float d = sigmoidf_(xj) - tg; cost += d * d; }
                else cost += tg * softplusf_(-xj) + (1.0f - tg) * softplusf_(xj);
            }
            if (!live) { m = 0.f; sv = 0.f; xj = 0.f; }          // keep pad columns zero
            const float ms = m * e.mean_scale;
            if (e.pre) e.pre[off] = xj;
            if (e.mean) e.mean[off] = ms;
            if (e.sample) e.sample[off] = sv;
            if (e.mean_planes) {
                unsigned short p1, p2, p3;
                split3(ms, p1, p2, p3);
                e.mean_planes[off] = p1; e.mean_planes[e.plane_stride + off] = p2; e.mean_planes[2 * e.plane_stride + off] = p3;
            }
            if (e.sample_plane) e.sample_plane[off] = (unsigned short)(__builtin_bit_cast(unsigned, sv) >> 16);
            if (live) csum += e.colsum_kind == 0 ? ms : (e.colsum_kind == 1 ? tg - m : tg - sv);
        }
    }
    if (e.colsum) e.colsum[(int64_t)(r0 >> 2) * e.ld + col] = csum;
}

template <int MODE = -1>
__device__ __forceinline__ void act_quad(const EpiArgs& e, float x0, float x1, float x2, float x3, int r0, int col, bool live, float& cost)
{
    // the targets requested first, under the Philox rounds -- loaded row by row inside the loop they were eight memory
    // round trips in a row
    float tg4[4];
    act_quad_targets(e, r0, col, live, tg4);
    act_quad_tg<MODE>(e, x0, x1, x2, x3, r0, col, live, cost, tg4);
}

// cost_slot: index of this tile's cost partial (default: the block index; a kernel whose reducer blocks are a
// subset of the grid passes the tile index instead)
template <int BM, int BN, int NT = 512>
__device__ __forceinline__ void fused_tile_epilogue(const EpiArgs& e, float* T, int m0, int n0, int cost_slot = -1)
{
    constexpr int LDT = BN + 8;
    const int c = threadIdx.x & (BN - 1), rg0 = threadIdx.x / BN;
    const int col = n0 + c;
    const bool live = col < e.cols, incol = col < (int)e.ld;
    const float bias = live ? e.bias[col] : 0.f;
    float cost = 0.f;
#pragma unroll 1
    for (int rg = rg0; rg < BM / 4; rg += NT / BN) {
        const int r0 = m0 + 4 * rg;
        if (r0 >= e.rows || !incol) continue;
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = T[(4 * rg + j) * LDT + c] + bias;
        act_quad(e, x[0], x[1], x[2], x[3], r0, col, live, cost);
    }
    if (e.cost_partials) {
        const float tot = block_sum(cost, T + BM * LDT);
        if (threadIdx.x == 0) e.cost_partials[cost_slot < 0 ? (int)blockIdx.x : cost_slot] = tot;
    }
}


// The same epilogue for a 128 x 64 tile on 512 threads in ONE pass: a thread owns 4 rows x 4 consecutive columns.  With
// only 8 waves on the CU (the GEMM's own) latency cannot be hidden by occupancy, so all global loads (the four target
// rows through the minibatch index) are issued up front, and every store is 8 or 16 bytes (one per plane and row, one
// float4 of bias statistics) instead of 2-byte plane stores per element.  Same arithmetic per element as act_quad (same
// Philox words: one block per column and 4-row group), so samples are bit-identical to the two-launch path.
// GAUSS / SAMPLE are the run-time facts e.gauss and (e.sample || e.sample_plane) as COMPILE-time ones (the caller branches
// once): with every case in one body the arithmetic of a 4 x 4 block took 6.4 us on the CU's eight waves (Philox, Box-Muller
// and both activations compiled in, few registers left to overlap anything: phase stamps, profiles/r03zr_…).
template <int NT, bool GAUSS, bool SAMPLE>
__device__ __forceinline__ void fused_tile_epilogue_4x4_t(const EpiArgs& e, float* T, int m0, int n0,
                                                          unsigned long long* stamps /* diagnostic builds: 3 slots */)
{
    constexpr int BM = 128, BN = 64, LDT = BN + 8;
    static_assert(NT == 512, "one 4 x 4 block per thread");
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;       // column quad 0..15, row group 0..31
    const int col0 = n0 + 4 * cq, r0 = m0 + 4 * rg;
    float cost = 0.f;
    if (r0 < e.rows && col0 < (int)e.ld) {
        float4 tg[4];
        const bool want_tg = e.target != nullptr;
        if (want_tg) {
            int64_t srow[4];
            epi_target_rows4(e, r0, srow);
#pragma unroll
            for (int j = 0; j < 4; ++j) tg[j] = *reinterpret_cast<const float4*>(e.target + srow[j] * e.ld_target + col0);
        }
        const float4 b4 = make_float4(col0 < e.cols ? e.bias[col0] : 0.f, col0 + 1 < e.cols ? e.bias[col0 + 1] : 0.f,
                                      col0 + 2 < e.cols ? e.bias[col0 + 2] : 0.f, col0 + 3 < e.cols ? e.bias[col0 + 3] : 0.f);
        float4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = *reinterpret_cast<const float4*>(T + (4 * rg + j) * LDT + 4 * cq);
            x[j].x += b4.x; x[j].y += b4.y; x[j].z += b4.z; x[j].w += b4.w;
        }
        constexpr bool need_u = SAMPLE, need_z = SAMPLE && GAUSS;
        float4 ms[4], sv[4], pre[4];
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (stamps && threadIdx.x == 0) {      // targets, bias and the parked rows have arrived
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            stamps[0] = wall_clock64();
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = col0 + c;
            const bool live = col < e.cols;
            uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
            if constexpr (need_u) {
                const uint64_t g0 = e.rng.row_offset + (uint64_t)r0;
                philox_rows4(e.rng, e.rng.draw, g0, (uint32_t)col, wa);
                if constexpr (need_z) philox_rows4(e.rng, e.rng.draw | MDBN_NORMAL_BIT, g0, (uint32_t)col, wb);
            }
            float csum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xj = comp(x[j], c);
                float m, s1 = 0.f;
                if constexpr (GAUSS) {
                    m = xj;
                    if constexpr (need_u) {
                        const float u1 = philox_u01(wa[j]), u2 = philox_u01(wb[j]);
                        s1 = m + sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    }
                } else {
                    m = sigmoidf_(xj);
                    if constexpr (need_u) s1 = philox_u01(wa[j]) < m ? 1.0f : 0.0f;
                }
                float t1 = 0.f;
                if (want_tg && live && r0 + j < e.rows) {
                    t1 = comp(tg[j], c);
                    if constexpr (GAUSS) { const float d = sigmoidf_(xj) - t1; cost += d * d; }
                    else cost += t1 * softplusf_(-xj) + (1.0f - t1) * softplusf_(xj);
                }
                if (!live) { m = 0.f; s1 = 0.f; xj = 0.f; }
                const float m1 = m * e.mean_scale;
                setc(ms[j], c, m1); setc(sv[j], c, s1); setc(pre[j], c, xj);
                if (live && r0 + j < e.rows) csum += e.colsum_kind == 0 ? m1 : (e.colsum_kind == 1 ? t1 - m : t1 - s1);
            }
            setc(cs, c, csum);
        }
        if (stamps && threadIdx.x == 0) { asm volatile("" :: "v"(cs.x), "v"(ms[3].w)); stamps[1] = wall_clock64(); }     // arithmetic done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (r0 + j >= e.rows) break;
            const int64_t off = (int64_t)(r0 + j) * e.ld + col0;
            if (e.pre) *reinterpret_cast<float4*>(e.pre + off) = pre[j];
            if (e.mean) *reinterpret_cast<float4*>(e.mean + off) = ms[j];
            if (e.sample) *reinterpret_cast<float4*>(e.sample + off) = sv[j];
            if (e.mean_planes) store_planes4(e.mean_planes, e.plane_stride, off, ms[j]);
            if (e.sample_plane) {
                uint2 w;
                w.x = (__builtin_bit_cast(unsigned, sv[j].x) >> 16) | (__builtin_bit_cast(unsigned, sv[j].y) & 0xffff0000u);
                w.y = (__builtin_bit_cast(unsigned, sv[j].z) >> 16) | (__builtin_bit_cast(unsigned, sv[j].w) & 0xffff0000u);
                *reinterpret_cast<uint2*>(e.sample_plane + off) = w;
            }
        }
        if (e.colsum) *reinterpret_cast<float4*>(e.colsum + (int64_t)(r0 >> 2) * e.ld + col0) = cs;
        if (stamps && threadIdx.x == 0) {      // this wave's stores acknowledged
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamps[2] = wall_clock64();
        }
    }
    if (e.cost_partials) {
        const float tot = block_sum(cost, T + BM * LDT);
        if (threadIdx.x == 0) e.cost_partials[blockIdx.x] = tot;
    }
}

template <int NT = 512>
__device__ __forceinline__ void fused_tile_epilogue_4x4(const EpiArgs& e, float* T, int m0, int n0,
                                                        unsigned long long* stamps = nullptr)
{
    const bool sample = e.sample != nullptr || e.sample_plane != nullptr;
    if (e.gauss) {
        if (sample) fused_tile_epilogue_4x4_t<NT, true, true>(e, T, m0, n0, stamps);
        else fused_tile_epilogue_4x4_t<NT, true, false>(e, T, m0, n0, stamps);
    } else {
        if (sample) fused_tile_epilogue_4x4_t<NT, false, true>(e, T, m0, n0, stamps);
        else fused_tile_epilogue_4x4_t<NT, false, false>(e, T, m0, n0, stamps);
    }
}

// Parameter update applied by the statistics GEMM to the tile it just computed (parked in LDS, row
// stride BN + 8): W and W_speed are read and written once, S never touches HBM.  The GEMM reads
// only V2 / P2, so updating W in place under it is safe.  A thread owns one float4 column group and
// walks rows; 4 rows of loads are in flight per round.
#ifndef FUSED_UPD_RB
#define FUSED_UPD_RB 4      // rows of W / W_speed (/ W0) loads in flight per thread and round
#endif
template <int BM, int BN, int NT = 512>
__device__ __forceinline__ void fused_update_epilogue(const UpdEpi& u, const float* T, int m0, int n0)
{
    constexpr int LDT = BN + 8, C4 = BN / 4, RSTEP = NT / C4, RB = FUSED_UPD_RB;
    const int c4 = threadIdx.x % C4, rr = threadIdx.x / C4;
    const int col = n0 + 4 * c4;
    if (col >= (int)u.ld) return;
    const float two_lr_l1 = upd_two_lr_l1(u.lr, u.l1);
    const float decay = upd_decay(u.lr, u.l2);
#pragma unroll 1
    for (int r = rr; r < BM; r += RSTEP * RB) {
        // (loads unconditional at clamped rows, issued as one batch; only the stores are predicated: with `if (row < rows)
        //  load`, every row's loads sat in their own block behind a wait)
        float4 w[RB], sp[RB], w0[RB];
        const float* w0base = u.W0 ? u.W0 : u.W;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int rowc = min(m0 + min(r + b * RSTEP, BM - 1), u.rows - 1);
            const int64_t off = (int64_t)rowc * u.ld + col;
            w[b] = *reinterpret_cast<const float4*>(u.W + off);
            sp[b] = *reinterpret_cast<const float4*>(u.Ws + off);
            w0[b] = *reinterpret_cast<const float4*>(w0base + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = m0 + r + b * RSTEP;
            if (r + b * RSTEP < BM && row < u.rows) {
                const int64_t off = (int64_t)row * u.ld + col;
                const float4 st = *reinterpret_cast<const float4*>(T + (r + b * RSTEP) * LDT + 4 * c4);
                float4 wn, sn;
                update_rule4(w[b], sp[b], st, w0[b], u.inv_bs, u.wc, decay, u.l1, two_lr_l1, u.mu, u.lr, wn, sn);
                *reinterpret_cast<float4*>(u.W + off) = wn;
                *reinterpret_cast<float4*>(u.Ws + off) = sn;
                if (u.Wp) store_planes4(u.Wp, u.wp_stride, off, wn);
            }
        }
    }
}

}  // namespace mdbn
