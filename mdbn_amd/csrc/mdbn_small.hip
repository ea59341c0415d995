// One launch per CD-k step for layers whose weights fit ONE CU's LDS (gfx950 / MI355X).
//
// The reference trains many such layers (MDBN.py:45-52, AMLsm2.py:242-340: 512 -> 40, 400 -> 40, 200 -> 20, 100 -> 24 -> 3, the
// joint layer), each step a scan of k gibbs_hvh (rbm.py:318-336).  On the multi-launch path such a step is 2 k + 4 dependent
// launches that each sit on the 5-7 us floor of a dependent tiny kernel (512 -> 40, CD-5, B = 512: 14 launches, 128 us for
// 0.27 GFLOP).  Here W (f32, <= ~100 KB) is staged into LDS once per workgroup, and a workgroup runs the WHOLE chain for
// FOUR-row slabs of the minibatch -- gather, positive phase, k x (propdown, propup) with the fused activations and the Philox
// draws, the reconstruction cost, and its share of the statistics S = v0^T ph - nv^T nh, accumulated in registers over its
// slabs -- on the exact-f32 MFMA with every operand read from LDS.  It writes ONE partial [S | s_h | s_v | cost] per
// workgroup; a second, small launch (small_finish_kernel) sums the partials in a fixed order and applies the parameter
// update (or stores the statistics for a data-parallel all-reduce).  No workgroup waits for another.  Same Philox addressing
// as every other path (counter = (column, global row >> 2, draw, step)): the same uniforms meet probabilities that differ
// from the multi-launch path's by fp32 summation order only.
//
// Why four rows.  The chain of a slab is a string of dependent passes; its length is what a step costs, and with 16-row
// slabs on v_mfma_f32_16x16x4_f32 (round 4's first version: 59 us for 512 -> 40 CD-5 at B = 512 on 32 workgroups) every pass
// sat on the f32 matrix pipe of ONE CU: 3 072 cycles per pass, 7/8 of the chip idle.  v_mfma_f32_4x4x1_16b_f32 computes 16
// independent 4 x 4 x 1 blocks; with the A operand BROADCAST from one block to all (cbsz = 4, abid = u) an instruction is a
// rank-1 update of a 4-row x 64-column tile -- 4 rows are exactly one Philox block -- so a minibatch of 512 rows is 128
// workgroups and a pass costs a quarter of the MFMA time.  One A register (lane 4 b + i holds X[i][k0 + b]) carries the A
// operands of 16 k-steps: 16 MFMAs name it with abid = 0 .. 15 (probe of the layouts and the issue cost -- 9.5 cycles per
// MFMA and wave on two accumulator chains: scripts/experiments/mfma4x4_probe.hip).
//
// MFMA operand layout (4x4x1, 16 blocks): A[block b = lane >> 2][i = lane & 3], B[b][j = lane & 3], D[b][i = register e][j]:
// a lane's four accumulator registers are the FOUR ROWS of output column 4 b + j = lane -- one Philox block.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_small.h"

namespace mdbn {

typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
// Every LDS pointer of this file carries its address space in its TYPE: held as plain `float*` (in arrays, across lambdas)
// hipcc loses track of it and emits FLAT loads -- the vector-memory path with an aperture check, several times slower than
// ds_read and counted on vmcnt (seen in the ISA: v_lshl_add_u64 pointer arithmetic and s_waitcnt vmcnt(0) in the MFMA loops).
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) sf32x4 lds_f4;      // 16-byte LDS accesses (float4 is a class: no address-space copy)

// The barriers of this kernel order LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL store of the
// wave (the inspection copies and chain taps each epilogue writes): ~1-2 us per barrier, 26 barriers per CD-5 slab.
#define SM_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {

#ifdef MDBN_STAMP_CLK   // diagnostic builds (with -DMDBN_STAMP): shader-clock cycles workgroup 0 / wave 0 spends in the parts of a pass
                        // (slots 48..55); each costs a memory round trip of thread 0: the phase stamps are then ~0.1 us per part too long
__device__ unsigned long long* g_sm_clk = nullptr;
#define SM_CLK_BEGIN() const long long clk_ = clock64()
#define SM_CLK_ADD(SLOT) do { if (g_sm_clk && blockIdx.x == 0 && threadIdx.x == 0) g_sm_clk[SLOT] += (unsigned long long)(clock64() - clk_); } while (0)
#else
#define SM_CLK_BEGIN() do { } while (0)
#define SM_CLK_ADD(SLOT) do { } while (0)
#endif

__device__ __forceinline__ int64_t sm_src_row(const SmallCdArgs& a, int row)
{
    if (!a.idx) return row;
    int64_t s = a.idx64 ? reinterpret_cast<const int64_t*>(a.idx)[row] : (int64_t)reinterpret_cast<const int32_t*>(a.idx)[row];
    if (s < 0) s += a.n_data;
    return s < 0 ? 0 : (s >= a.n_data ? a.n_data - 1 : s);
}

#ifndef SM_RNG_WAVE
#define SM_RNG_WAVE 0       // 1: a ninth wave draws the propup epilogues' Philox blocks under the other waves' reduction loops
#endif
#ifndef SM_SUM_UNROLLED
#define SM_SUM_UNROLLED 1   // propup epilogue: all chunk partials requested at once (0: a rolled read-wait-add loop)
#endif

// 16 rank-1 updates of a 4 x 64 tile: A register `A_` (16 k-steps, one per block), B values B_(0) .. B_(15); two accumulator
// chains (even / odd k) so that an MFMA never waits for the one before it.  (abid must be an immediate: spelled out.)
#define SM_MMA1(ACC, A_, BV, U) ACC = __builtin_amdgcn_mfma_f32_4x4x1f32(A_, BV, ACC, 4, U, 0)
#define SM_MMA16(ACC0, ACC1, A_, B_)                                                                                   \
    SM_MMA1(ACC0, A_, B_(0), 0);   SM_MMA1(ACC1, A_, B_(1), 1);   SM_MMA1(ACC0, A_, B_(2), 2);   SM_MMA1(ACC1, A_, B_(3), 3);   \
    SM_MMA1(ACC0, A_, B_(4), 4);   SM_MMA1(ACC1, A_, B_(5), 5);   SM_MMA1(ACC0, A_, B_(6), 6);   SM_MMA1(ACC1, A_, B_(7), 7);   \
    SM_MMA1(ACC0, A_, B_(8), 8);   SM_MMA1(ACC1, A_, B_(9), 9);   SM_MMA1(ACC0, A_, B_(10), 10); SM_MMA1(ACC1, A_, B_(11), 11); \
    SM_MMA1(ACC0, A_, B_(12), 12); SM_MMA1(ACC1, A_, B_(13), 13); SM_MMA1(ACC0, A_, B_(14), 14); SM_MMA1(ACC1, A_, B_(15), 15)

// One pass of the chain, D[4][N] = A[4][K] * op(W), in two forms.  No operand masking anywhere: the pad columns of the
// 4-row buffers and the pad rows / columns of W's image hold exact zeros (every writer keeps them so), a lane's column index
// is clamped into the image, and a column n >= N computes something finite that `epi` discards.
//
// sm_up (propup, K = V long, N = H: one or two 64-column tiles): work items = (tile, K chunk) dealt over the waves; a lane
// reads W[k][its column] (lanes side by side: no conflicts), 16 k-steps per A register, the next group's operands in flight
// under the MFMAs of this one.  The chunk partials go through `part` as one float4 per (chunk, column); after a barrier
// thread c sums the chunks of column c in chunk order and applies `epi` ONCE.
//
// sm_down (propdown, K = H short, N = V: up to 8 tiles, one per wave): a lane reads 16 bytes of ITS row of W (its output
// column) per 4 k-steps; the wave applies `epi` to its accumulator registers.
struct UpFrag { float a; float b[16]; };

// LDW: W's LDS pitch as a compile-time constant (small_layout hands out 20 / 44 / 68 / 132 for H <= 132) -- the 16 reads of a
// group are then one base register + immediate offsets; with a run-time pitch (LDW = 0) each read costs an address add, and
// the loop is bound by its instruction count (57 instead of 34 per 16 MFMAs: 0.96 us per pass at 512 -> 40, stamped).
template <int LDW>
__device__ __forceinline__ void sm_up_loop(lds_cf* X, lds_cf* Wl, const SmallLayout& L, lds_f* part, int wave, int lane)
{
    const int bi = (lane & 3) * L.ldx + (lane >> 2);
    const int ldw = LDW ? LDW : L.ldw;
    for (int item = wave; item < L.tiles_up * L.ks_up; item += SM_NW) {
        SM_CLK_BEGIN();
        const int tile = item % L.tiles_up, ch = item / L.tiles_up;
        const int k0 = ch * L.per_up;
        const int groups = (min(L.Vp, k0 + L.per_up) - k0) >> 4;
        lds_cf* ap = X + bi + k0;
        lds_cf* bp = Wl + k0 * ldw + min(64 * tile + lane, ldw - 1);
        sf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        auto load = [&](UpFrag& f, int g) {
            f.a = ap[16 * g];
            lds_cf* q = bp + 16 * g * ldw;
#pragma unroll
            for (int u = 0; u < 16; ++u) f.b[u] = q[u * ldw];
        };
        UpFrag f0, f1;
        load(f0, 0);
        for (int g = 0; g + 2 <= groups; g += 2) {       // (scheduling barriers: hipcc otherwise sinks each load to its first use)
            load(f1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
#define SM_B(U) f0.b[U]
            SM_MMA16(acc0, acc1, f0.a, SM_B);
#undef SM_B
            __builtin_amdgcn_sched_barrier(0);
            load(f0, min(g + 2, groups - 1));
            __builtin_amdgcn_sched_barrier(0);
#define SM_B(U) f1.b[U]
            SM_MMA16(acc0, acc1, f1.a, SM_B);
#undef SM_B
            __builtin_amdgcn_sched_barrier(0);
        }
        if (groups & 1) {
#define SM_B(U) f0.b[U]
            SM_MMA16(acc0, acc1, f0.a, SM_B);
#undef SM_B
        }
        *(lds_f4*)(part + 4 * (ch * L.H64 + 64 * tile + lane)) = acc0 + acc1;
        SM_CLK_ADD(48);
    }
}

// `rng` (the ninth wave's job, under the other waves' reduction loops): the Philox blocks the epilogue will need, one per
// column, into `U` -- 10 rounds of quarter-rate multiplies are half of the epilogue's ~1 850 cycles, and they depend on
// nothing the pass computes.
template <class Rng, class Epi>
__device__ __forceinline__ void sm_up(lds_cf* X, lds_cf* Wl, const SmallLayout& L, lds_f* part, int wave, int lane, Rng&& rng, Epi&& epi)
{
    if (SM_RNG_WAVE && wave == SM_NW) rng();
    else
    switch (L.ldw) {
        case 20: sm_up_loop<20>(X, Wl, L, part, wave, lane); break;
        case 44: sm_up_loop<44>(X, Wl, L, part, wave, lane); break;
        case 68: sm_up_loop<68>(X, Wl, L, part, wave, lane); break;
        case 132: sm_up_loop<132>(X, Wl, L, part, wave, lane); break;
        default: sm_up_loop<0>(X, Wl, L, part, wave, lane); break;
    }
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(49); }
    SM_CLK_BEGIN();
    for (int col = threadIdx.x; col < L.H64; col += SM_NT + 64 * SM_RNG_WAVE) {
        // (four chunk partials requested at once, summed in chunk order: a rolled read-wait-add loop is an LDS round trip per
        //  chunk; chunks past ks_up re-read the last one and add an exact zero)
#if SM_SUM_UNROLLED
        sf32x4 pc[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) pc[ch] = *(const lds_f4*)(part + 4 * (min(ch, L.ks_up - 1) * L.H64 + col));
        sf32x4 x = pc[0];
#pragma unroll
        for (int ch = 1; ch < 4; ++ch) x += ch < L.ks_up ? pc[ch] : sf32x4{0.f, 0.f, 0.f, 0.f};
        if (L.ks_up > 4) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) pc[ch] = *(const lds_f4*)(part + 4 * (min(4 + ch, L.ks_up - 1) * L.H64 + col));
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) x += 4 + ch < L.ks_up ? pc[ch] : sf32x4{0.f, 0.f, 0.f, 0.f};
        }
#else
        sf32x4 x = *(const lds_f4*)(part + 4 * col);
        for (int ch = 1; ch < L.ks_up; ++ch) x += *(const lds_f4*)(part + 4 * (ch * L.H64 + col));
#endif
        epi(x, col);
    }
    SM_CLK_ADD(50);
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(51); }
}

template <class Epi>
__device__ __forceinline__ void sm_down(lds_cf* Hs, lds_cf* Wl, const SmallLayout& L, int wave, int lane, Epi&& epi)
{
    const int bi = (lane & 3) * L.ldhs + (lane >> 2);
    const int groups = L.Hp >> 4;
    for (int tile = wave; tile < L.tiles_dn; tile += SM_NW) {
        SM_CLK_BEGIN();
        lds_cf* ap = Hs + bi;
        const lds_f4* bp = (const lds_f4*)(Wl + min(64 * tile + lane, L.Vp - 1) * L.ldw);
        sf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        struct Frag { float a; sf32x4 b[4]; };
        auto load = [&](Frag& f, int g) {
            f.a = ap[16 * g];
#pragma unroll
            for (int q = 0; q < 4; ++q) f.b[q] = bp[4 * g + q];
        };
        Frag f0, f1;
        load(f0, 0);
        for (int g = 0; g + 2 <= groups; g += 2) {
            load(f1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
#define SM_B(U) f0.b[(U) >> 2][(U) & 3]
            SM_MMA16(acc0, acc1, f0.a, SM_B);
#undef SM_B
            __builtin_amdgcn_sched_barrier(0);
            load(f0, min(g + 2, groups - 1));
            __builtin_amdgcn_sched_barrier(0);
#define SM_B(U) f1.b[(U) >> 2][(U) & 3]
            SM_MMA16(acc0, acc1, f1.a, SM_B);
#undef SM_B
            __builtin_amdgcn_sched_barrier(0);
        }
        if (groups & 1) {
#define SM_B(U) f0.b[(U) >> 2][(U) & 3]
            SM_MMA16(acc0, acc1, f0.a, SM_B);
#undef SM_B
        }
        SM_CLK_ADD(52);
        epi(acc0 + acc1, 64 * tile + lane);
        SM_CLK_ADD(53);
    }
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(54); }
}

// S tile += X^T M over the slab's 4 rows, for the positive and the negative pair at once.  Work item = (64 rows of S, 64
// columns of S), one per wave (small_shape_ok: at most 8 items).  Here the HIDDEN means are the broadcast operand (A register:
// lane l holds M[r][64 th + l], block u = hidden columns 4 u .. 4 u + 3) and the visible row is B (lane l = visible unit
// 64 tv + l), so accumulator u of lane l is S[64 tv + l][64 th + 4 u .. + 3]: 16 bytes of ONE row of S, stored as such.
__device__ __forceinline__ void sm_stats(sf32x4 (&accS)[SM_MAXQ], lds_cf* X0, lds_cf* M0, lds_cf* Xn, lds_cf* Mn,
                                         const SmallLayout& L, int nq, int wave, int lane)
{
    if (wave >= L.tiles_dn * L.tiles_up) return;
    const int tv = wave % L.tiles_dn, th = wave / L.tiles_dn;
    float a[8], b[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = M0[r * L.ldhs + 64 * th + lane]; b[r] = X0[r * L.ldx + 64 * tv + lane];
        a[4 + r] = Mn[r * L.ldhs + 64 * th + lane]; b[4 + r] = Xn[r * L.ldx + 64 * tv + lane];
    }
    // (two accumulators per block of code: a lone chain of dependent 4x4x1 MFMAs issues every 13.7 cycles, two every 9.5)
#define SM_STAT(U)                                                                                                     \
    if ((U) + 1 < nq) {                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                                \
            accS[U] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], b[r], accS[U], 4, U, 0);                                \
            accS[(U) + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], b[r], accS[(U) + 1], 4, (U) + 1, 0);              \
        }                                                                                                              \
    } else if ((U) < nq) {                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) accS[U] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], b[r], accS[U], 4, U, 0); \
    }
    SM_STAT(0) SM_STAT(2) SM_STAT(4) SM_STAT(6) SM_STAT(8) SM_STAT(10) SM_STAT(12) SM_STAT(14)
#undef SM_STAT
}

// what the passes of one slab share (plain pointers and sizes: copied into registers once)
struct SmCtx {
    lds_f *Wl, *X0, *Xa, *Xb, *Hs, *M0, *Mn, *part, *csP, *csN, *csV, *hbl, *vbl;
    __attribute__((address_space(3))) uint32_t* U;      // [H64][4] Philox words of the running propup's epilogue
    int row0; uint64_t grow0;
};

// v_t | h_{t-1}: RBM sigmoid + Bernoulli (draw 2t-1), GRBM linear mean (rbm.py:647-660, error_free).  LAST: the chain's last
// step also yields the reconstruction cost (rbm.py:372-374,449-482; GRBM :690-699: a sigmoid is applied to the linear
// mean) against v0 (still in LDS: X0), the column sums of v0 - nv, and (RBM) the visible MEAN for the statistics.  TAPS:
// inspection copies / chain taps.  The flags are compile-time: a pass executes only the instructions it needs (at ~5 cycles
// per instruction and wave a generic epilogue costs more than the pass's MFMAs).
template <bool GAUSS, bool LAST, bool TAPS>
__device__ __forceinline__ void sm_step_down(const SmallCdArgs& a, const SmCtx& c, int t, float& cost, int wave, int lane)
{
    const SmallLayout& L = a.L;
    const int V = a.V, B = a.B;
    const int64_t ldv = a.ldv;
    sm_down(c.Hs, c.Wl, L, wave, lane,
            [&](const sf32x4& x, int col) {
                // (every LDS read first, every LDS write last: hipcc cannot tell the buffers apart and would otherwise
                // serialise a read behind each write -- a round trip per element)
                const bool live = col < V;
                const float bias = c.vbl[col];
                float tg[4] = {0.f, 0.f, 0.f, 0.f};
                if (LAST) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) tg[e] = c.X0[e * L.ldx + col];
                }
                const float cs0 = LAST ? c.csV[col] : 0.f;
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (!GAUSS) philox_rows4(a.rng, (uint32_t)(2 * t - 1), c.grow0, (uint32_t)col, w);
                float m[4], sv[4], cs = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = live && c.row0 + e < B;
                    const float pre = x[e] + bias;
                    sv[e] = 0.f;
                    if (GAUSS) m[e] = ok ? pre : 0.f;
                    else {
                        m[e] = ok ? sigmoidf_(pre) : 0.f;
                        sv[e] = ok && philox_u01(w[e]) < m[e] ? 1.0f : 0.0f;
                    }
                    if (LAST && ok) {
                        if (GAUSS) { const float d = sigmoidf_(pre) - tg[e]; cost += d * d; }
                        else cost += tg[e] * softplusf_(-pre) + (1.0f - tg[e]) * softplusf_(pre);
                        cs += tg[e] - m[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c.Xa[e * L.ldx + col] = GAUSS ? m[e] : sv[e];
                    if (!GAUSS && LAST) c.Xb[e * L.ldx + col] = m[e];
                }
                if (LAST) c.csV[col] = cs0 + cs;        // (a column belongs to one lane)
                if (TAPS && col < (int)ldv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = c.row0 + e;
                        if (row < B) {
                            if (LAST && a.keep) a.V2[(int64_t)(B + row) * ldv + col] = m[e];
                            if (!GAUSS) {
                                if (a.keep) a.vs[(int64_t)row * ldv + col] = sv[e];
                                if (a.trace_v) a.trace_v[((int64_t)(t - 1) * B + row) * ldv + col] = sv[e];
                            }
                        }
                    }
                }
            });
}

// h_t | v_t (t = 0: from v0): from the mean for GRBM (rbm.py:669), from the sample for RBM (rbm.py:246).  KIND 0: the
// positive phase (mean kept for the statistics + sample, draw 0); 1: a middle step (sample only, draw 2t); 2: the chain's
// end (-nh for the statistics, no sample: CD does not materialise it).
template <int KIND, bool TAPS>
__device__ __forceinline__ void sm_step_up(const SmallCdArgs& a, const SmCtx& c, lds_cf* X, int t, int wave, int lane)
{
    const SmallLayout& L = a.L;
    const int H = a.H, B = a.B;
    const int64_t ldh = a.ldh;
    typedef __attribute__((address_space(3))) su32x4 lds_u4;
    sm_up(X, c.Wl, L, c.part, wave, lane,
          [&]() {
              if (KIND != 2) {
                  for (int col = lane; col < L.H64; col += 64) {
                      uint32_t w[4];
                      philox_rows4(a.rng, (uint32_t)(2 * t), c.grow0, (uint32_t)col, w);
                      *(lds_u4*)(c.U + 4 * col) = su32x4{w[0], w[1], w[2], w[3]};
                  }
              }
          },
          [&](const sf32x4& x, int col) {
              const bool live = col < H;
              const float bias = c.hbl[col];
              lds_f* cd = (KIND == 2 ? c.csN : c.csP) + col;
              const float cs0 = KIND != 1 ? *cd : 0.f;
              su32x4 w = {0u, 0u, 0u, 0u};
              if (KIND != 2) {
                  if (SM_RNG_WAVE) w = *(const lds_u4*)(c.U + 4 * col);
                  else {
                      uint32_t w4[4];
                      philox_rows4(a.rng, (uint32_t)(2 * t), c.grow0, (uint32_t)col, w4);
                      w = su32x4{w4[0], w4[1], w4[2], w4[3]};
                  }
              }
              float m[4], sv[4], cs = 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                  const bool ok = live && c.row0 + e < B;
                  const float p = ok ? sigmoidf_(x[e] + bias) : 0.f;
                  m[e] = KIND == 2 ? -p : p;
                  cs += m[e];
                  sv[e] = (KIND != 2 && ok && philox_u01(w[e]) < p) ? 1.0f : 0.0f;
              }
              if (KIND != 1) {
                  lds_f* Ml = KIND == 2 ? c.Mn : c.M0;
#pragma unroll
                  for (int e = 0; e < 4; ++e) Ml[e * L.ldhs + col] = m[e];
                  *cd = cs0 + cs;
              }
              if (KIND != 2) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) c.Hs[e * L.ldhs + col] = sv[e];
              }
              if (TAPS && col < (int)ldh) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                      const int row = c.row0 + e;
                      if (row < B) {
                          if (KIND != 1 && a.keep) a.P2[(int64_t)((KIND == 2 ? B : 0) + row) * ldh + col] = m[e];
                          if (KIND != 2) {
                              if (a.keep) a.hs[(int64_t)row * ldh + col] = sv[e];
                              if (a.trace_h) a.trace_h[((int64_t)t * B + row) * ldh + col] = sv[e];
                          }
                      }
                  }
              }
          });
}

}  // namespace

template <bool GAUSS, bool TAPS>
__global__ __launch_bounds__(SM_NT + 64 * SM_RNG_WAVE) void small_cd_kernel(SmallCdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const SmallLayout& L = a.L;
    lds_f* const lds = (lds_f*)sm;
    SmCtx c;
    c.Wl = lds + L.oW;
    c.X0 = lds + L.oX0;             // v0: operand of the positive phase, target of the reconstruction cost, statistics
    c.Xa = lds + L.oXa;             // visible operand of the next propup (GRBM: nv mean; RBM: v sample)
    c.Xb = lds + L.oXb;             // RBM, last Gibbs step: nv MEAN (the statistics use the mean, the chain the sample)
    c.Hs = lds + L.oHs;             // hidden sample (operand of the next propdown)
    c.M0 = lds + L.oM0;             // ph
    c.Mn = lds + L.oMn;             // -nh
    c.part = lds + L.oPart;
    c.csP = lds + L.oCsP;           // [H64] column sums of  ph over this workgroup's slabs
    c.csN = lds + L.oCsN;           //                       -nh
    c.csV = lds + L.oCsV;           // [V64] column sums of v0 - nv
    c.hbl = lds + L.oHb;
    c.vbl = lds + L.oVb;
    c.U = (__attribute__((address_space(3))) uint32_t*)(lds + L.oU);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool worker = wave < SM_NW;          // waves 0..7 run the passes; wave 8 draws the random numbers ahead of them (sm_up)
    const int V = a.V, H = a.H, B = a.B;
    const int64_t ldv = a.ldv, ldh = a.ldh;
    const int nslabs = (B + SM_ROWS - 1) / SM_ROWS;
    const int k_steps = a.k;
    int n_stamp = 0;
    (void)n_stamp;
#ifdef MDBN_STAMP       // diagnostic builds (scripts/experiments/small_stamps.py): wall-clock stamps of workgroup 0's phases
    const long long sclk0 = clock64();
#ifdef MDBN_STAMP_CLK
    if (blockIdx.x == 0 && tid == 0) { g_sm_clk = a.stamps; if (a.stamps) for (int i = 48; i < 56; ++i) a.stamps[i] = 0; }
#endif
#define SM_STAMP() do { if (a.stamps && blockIdx.x == 0 && tid == 0 && n_stamp < 46) a.stamps[1 + n_stamp++] = wall_clock64(); } while (0)
#else
#define SM_STAMP() do { } while (0)
#endif
    SM_STAMP();

    // ---- every first-touch load of the workgroup in ONE burst (a round trip to memory another XCD just wrote is ~2 us, and
    //      four dependent ones -- W, biases, index, rows -- were 3.6 us of a 17-us kernel): the first slab's source-row indices
    //      first, the biases and the first batch of W behind them, then the slab's rows as soon as the indices are back
    //      (vector-memory loads return in order: waiting for the oldest does not wait for the rest).
    // the slab's rows: ONE 16-byte piece per thread (4 rows x ldv / 4 pieces <= 512: small_ld_ok); the pad columns of X0 are
    // zeroed once, here -- nobody else writes them
    const int q4x = L.ldx >> 2, dq4 = (int)(ldv >> 2);
    const int g_r = tid / dq4, g_c4 = tid - g_r * dq4;                    // this thread's piece: (row of the slab, piece of the row)
    const bool g_mine = worker && g_r < SM_ROWS;
    {
        const int npad = q4x - dq4;
        if (worker && tid < SM_ROWS * npad) {
            const int r = tid / npad, cp = dq4 + (tid - r * npad);
            *(lds_f4*)(c.X0 + r * L.ldx + 4 * cp) = sf32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int row0_first = (int)blockIdx.x * SM_ROWS;
    int64_t srow0 = 0;
    if (g_mine && row0_first + g_r < B) srow0 = sm_src_row(a, row0_first + g_r);
    const float hb_r = tid < H ? a.hbias[tid] : 0.f, vb_r = tid < V ? a.vbias[tid] : 0.f;       // (H, V <= 512 = SM_NT)
    sf32x4 xg = {0.f, 0.f, 0.f, 0.f};
    if (worker) {
        // W image [Vp][ldw]: rows >= V and columns >= ldh zero (the pad columns of W below ldh are zero in memory).  Batches
        // of SM_WB loads per thread in flight, then their LDS stores.
        const int q4w = L.ldw >> 2, q4 = (int)(ldh >> 2);
        const int total = L.Vp * q4w + 4;                            // (+ the slack behind the last row)
        const int dr = SM_NT / q4w, dc = SM_NT - dr * q4w;
        int r = tid / q4w, c4 = tid - r * q4w;
        constexpr int SM_WB = 12;
        for (int e0 = tid; e0 < total || e0 == tid; e0 += SM_WB * SM_NT) {
            sf32x4 v[SM_WB];
#pragma unroll
            for (int u = 0; u < SM_WB; ++u) {
                v[u] = sf32x4{0.f, 0.f, 0.f, 0.f};
                if (r < V && c4 < q4) v[u] = *reinterpret_cast<const sf32x4*>(a.W + (int64_t)r * ldh + 4 * c4);
                r += dr; c4 += dc;
                if (c4 >= q4w) { c4 -= q4w; ++r; }
            }
            if (e0 == tid) {                                         // the first slab's rows ride behind the first batch
                __builtin_amdgcn_sched_barrier(0);
                if (g_mine && row0_first + g_r < B) xg = *reinterpret_cast<const sf32x4*>(a.data + srow0 * a.ld_data + 4 * g_c4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SM_WB; ++u)
                if (e0 + u * SM_NT < total) *(lds_f4*)(c.Wl + 4 * (e0 + u * SM_NT)) = v[u];
        }
    }
    if (tid < L.H64) { c.hbl[tid] = hb_r; c.csP[tid] = 0.f; c.csN[tid] = 0.f; }
    if (tid < L.V64) { c.vbl[tid] = vb_r; c.csV[tid] = 0.f; }
    sf32x4 accS[SM_MAXQ];
#pragma unroll
    for (int u = 0; u < SM_MAXQ; ++u) accS[u] = sf32x4{0.f, 0.f, 0.f, 0.f};
    float cost = 0.f;
    // 4-column groups of S this wave's statistics tile holds (the columns up to ldh are stored: pads as exact zeros)
    const int th_w = wave / L.tiles_dn, tv_w = wave - th_w * L.tiles_dn;
    const int nq = max(0, min(SM_MAXQ, ((int)ldh - 64 * th_w) >> 2));

    for (int slab = blockIdx.x; slab < nslabs; slab += gridDim.x) {
        c.row0 = slab * SM_ROWS;
        c.grow0 = a.rng.row_offset + (uint64_t)c.row0;                 // global row of the slab's first row (Philox address)
        if (slab != (int)blockIdx.x) SM_SYNC();                        // (the previous slab's last readers are done)
        SM_STAMP();
        // ---- x = train_set_x[indexes] (dbn.py:307): 4 rows into LDS (every thread resolves its own source rows; the first
        //      slab's are already in registers)
        if (slab != (int)blockIdx.x) {
            xg = sf32x4{0.f, 0.f, 0.f, 0.f};
            if (g_mine && c.row0 + g_r < B)
                xg = *reinterpret_cast<const sf32x4*>(a.data + sm_src_row(a, c.row0 + g_r) * a.ld_data + 4 * g_c4);
        }
        if (g_mine) {
            *(lds_f4*)(c.X0 + g_r * L.ldx + 4 * g_c4) = xg;
            if (TAPS && a.keep && c.row0 + g_r < B) *reinterpret_cast<sf32x4*>(a.V2 + (int64_t)(c.row0 + g_r) * ldv + 4 * g_c4) = xg;
        }
        SM_SYNC();
        SM_STAMP();
        // ---- positive phase (rbm.py:303)
        sm_step_up<0, TAPS>(a, c, c.X0, 0, wave, lane);
        SM_STAMP();
        // ---- k x gibbs_hvh (rbm.py:242-248, GRBM :662-671)
        for (int t = 1; t < k_steps; ++t) {
            sm_step_down<GAUSS, false, TAPS>(a, c, t, cost, wave, lane);
            SM_STAMP();
            sm_step_up<1, TAPS>(a, c, c.Xa, t, wave, lane);
            SM_STAMP();
        }
        sm_step_down<GAUSS, true, TAPS>(a, c, k_steps, cost, wave, lane);
        SM_STAMP();
        sm_step_up<2, TAPS>(a, c, c.Xa, k_steps, wave, lane);
        SM_STAMP();
        // ---- S += v0^T ph + nv_mean^T (-nh_mean)
        sm_stats(accS, c.X0, c.M0, GAUSS ? c.Xa : c.Xb, c.Mn, L, nq, wave, lane);
        SM_STAMP();
    }

    // ---- this workgroup's partials: the cost (one per wave: no barrier), a lane's 16-byte pieces of S, the column sums
    {
        float tot = cost;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (lane == 0 && worker) a.cost_partials[(int)blockIdx.x * SM_NW + wave] = tot;
    }
    // (layout of a partial: the 16-byte pieces in the order the lanes hold them -- [hidden tile th][visible tile tv][piece u]
    //  [lane] -- so that every store instruction writes 1 KB of consecutive memory; stored by row of S, 160 bytes apart per
    //  lane, the 5 120 scattered pieces of a workgroup took 2.5 us (stamped).  small_finish_kernel decodes the position.)
    if (wave < L.tiles_dn * L.tiles_up && 64 * tv_w + lane < V) {       // (pad lanes: rows of S that do not exist -- not stored, not read)
        sf32x4* Sp = reinterpret_cast<sf32x4*>(a.part_S) + (int64_t)blockIdx.x * small_part_quads(L, (int)ldh)
                     + (int64_t)(16 * th_w * L.tiles_dn + nq * tv_w) * 64 + lane;
        // non-temporal stores: nobody on this XCD reads the partial, and 10.5 MB of dirty L2 lines at the kernel's end cost 4.5 us
        // at 512 -> 40 (measured by not storing them), streamed 3.2 (profiles/r04z_small_partial_store_ab.log)
#define SM_PUT(U) if ((U) < nq) __builtin_nontemporal_store(accS[U], Sp + (U) * 64);
        SM_PUT(0) SM_PUT(1) SM_PUT(2) SM_PUT(3) SM_PUT(4) SM_PUT(5) SM_PUT(6) SM_PUT(7)
        SM_PUT(8) SM_PUT(9) SM_PUT(10) SM_PUT(11) SM_PUT(12) SM_PUT(13) SM_PUT(14) SM_PUT(15)
#undef SM_PUT
    }
    for (int j = tid; j < (int)ldh && worker; j += SM_NT) {
        a.posP[(int64_t)blockIdx.x * ldh + j] = j < L.H64 ? c.csP[j] : 0.f;
        a.negP[(int64_t)blockIdx.x * ldh + j] = j < L.H64 ? c.csN[j] : 0.f;
    }
    for (int i = tid; i < (int)ldv && worker; i += SM_NT) a.partV[(int64_t)blockIdx.x * ldv + i] = i < L.V64 ? c.csV[i] : 0.f;
    SM_STAMP();
#ifdef MDBN_STAMP
    if (a.stamps && blockIdx.x == 0 && tid == 0) { a.stamps[0] = (unsigned long long)n_stamp; a.stamps[63] = (unsigned long long)(clock64() - sclk0); }
#endif
#undef SM_STAMP
}

int small_blocks(int64_t B)
{
    const int64_t nslabs = (B + SM_ROWS - 1) / SM_ROWS;
    return (int)std::min<int64_t>(nslabs, SM_MAX_BLOCKS);
}

bool small_shape_ok(int64_t B, int64_t V, int64_t H, int gauss)
{
    if (B < 1 || V < 1 || H < 1 || V > 512 || H > 512) return false;
    const SmallLayout L = small_layout((int)V, (int)H, gauss != 0);
    if (L.bytes > SM_MAX_LDS) return false;
    return L.tiles_dn * L.tiles_up <= SM_NW;        // one 64 x 64 tile of S per wave
}

bool small_ld_ok(int64_t V, int64_t H, int64_t ldv, int64_t ldh)
{
    const SmallLayout L = small_layout((int)V, (int)H, false);
    return ldh % 4 == 0 && ldv % 4 == 0 && ldh >= H && ldv >= V && ldh <= L.ldw && ldh <= L.H64 && ldv <= L.ldx &&
           ldv <= SM_NT;      // (4 rows x ldv / 4 pieces of a slab: one per thread)
}

hipError_t launch_small_cd(const SmallCdArgs& a, hipStream_t s)
{
    const SmallLayout L = small_layout(a.V, a.H, a.gauss != 0);
    if (!small_shape_ok(a.B, a.V, a.H, a.gauss) || !small_ld_ok(a.V, a.H, a.ldv, a.ldh)) return hipErrorInvalidValue;
    const bool taps = a.keep || a.trace_h || a.trace_v;
    const int variant = (a.gauss ? 2 : 0) | (taps ? 1 : 0);
    static bool attr_set[4] = {false, false, false, false};
    const void* kerns[4] = {reinterpret_cast<const void*>(small_cd_kernel<false, false>), reinterpret_cast<const void*>(small_cd_kernel<false, true>),
                            reinterpret_cast<const void*>(small_cd_kernel<true, false>), reinterpret_cast<const void*>(small_cd_kernel<true, true>)};
    if (!attr_set[variant]) {
        hipError_t e = hipFuncSetAttribute(kerns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, SM_MAX_LDS);
        if (e != hipSuccess) return e;
        attr_set[variant] = true;
    }
    const dim3 grid(small_blocks(a.B)), block(SM_NT + 64 * SM_RNG_WAVE);
    SmallCdArgs k = a;
    k.L = L;
    switch (variant) {
        case 0: hipLaunchKernelGGL((small_cd_kernel<false, false>), grid, block, L.bytes, s, k); break;
        case 1: hipLaunchKernelGGL((small_cd_kernel<false, true>), grid, block, L.bytes, s, k); break;
        case 2: hipLaunchKernelGGL((small_cd_kernel<true, false>), grid, block, L.bytes, s, k); break;
        default: hipLaunchKernelGGL((small_cd_kernel<true, true>), grid, block, L.bytes, s, k); break;
    }
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Second launch: S = sum of the workgroups' partials, then either the parameter update of rbm.py:347-365 on it (single
// device: update_rule4, as update_kernel) or a plain store into the statistics buffer (data-parallel: the all-reduce
// follows); the trailing blocks run the finalize units (bias statistics, cost, bias half of the update).  `lanes` threads
// share one float4 of S: each sums a contiguous run of the partials in workgroup order (8 loads in flight), the runs are
// combined by a butterfly -- a fixed tree, the same in every lane -- so up to 128 partials cost two rounds of loads.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_finish_kernel(SmallFinArgs f)
{
    // (the arguments needed before the first load, requested in one batch: see gemm_planes_kernel)
    asm volatile("" :: "s"(f.part), "s"(f.nparts), "s"(f.n4p), "s"(f.V), "s"(f.q4), "s"(f.tiles_dn), "s"(f.lanes), "s"(f.do_upd),
                 "s"(f.upd.W), "s"(f.upd.Ws), "s"(f.upd.W0), "s"((int)gridDim.x));
    const int G = f.lanes, ipb = 256 / G;
    const int nbw = (int)((f.n4p + ipb - 1) / ipb);
    if ((int)blockIdx.x < nbw) {
        const int sub = threadIdx.x & (G - 1);
        const int64_t p = (int64_t)blockIdx.x * ipb + threadIdx.x / G;      // position in a partial
        // position -> (row of S, 4-column group): [th][tv][u][lane], nq(th) pieces per lane (small_cd_kernel)
        const int q = (int)(p >> 6), lane = (int)(p & 63);
        const int th = q / (16 * f.tiles_dn);
        const int nq = max(1, min(16, f.q4 - 16 * th));
        const int rem = q - 16 * th * f.tiles_dn;
        const int tv = rem / nq, u = rem - tv * nq;
        const int v = 64 * tv + lane;
        const bool in = p < f.n4p && v < f.V;
        const int per = (f.nparts + G - 1) / G;
        const int pb = sub * per, pe = min(f.nparts, pb + per);
        const float4* P = reinterpret_cast<const float4*>(f.part);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = pb; p0 < pe; p0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (in && p0 + u < pe) ? P[p + (int64_t)(p0 + u) * f.n4p] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }
        }
        for (int off = 1; off < G; off <<= 1) {
            s4.x += __shfl_xor(s4.x, off, 64); s4.y += __shfl_xor(s4.y, off, 64);
            s4.z += __shfl_xor(s4.z, off, 64); s4.w += __shfl_xor(s4.w, off, 64);
        }
        if (!in || sub != 0) return;
        const int64_t i = (int64_t)v * f.q4 + 16 * th + u;
        if (!f.do_upd) {
            reinterpret_cast<float4*>(f.S_out)[i] = s4;
            return;
        }
        const UpdEpi& up = f.upd;
        const float4 w = reinterpret_cast<const float4*>(up.W)[i], sp = reinterpret_cast<const float4*>(up.Ws)[i];
        const float4 w0 = up.W0 ? reinterpret_cast<const float4*>(up.W0)[i] : w;
        float4 wn, sn;
        update_rule4(w, sp, s4, w0, up.inv_bs, up.wc, upd_decay(up.lr, up.l2), up.l1, upd_two_lr_l1(up.lr, up.l1), up.mu, up.lr, wn, sn);
        reinterpret_cast<float4*>(up.W)[i] = wn;
        reinterpret_cast<float4*>(up.Ws)[i] = sn;
        if (up.Wp) store_planes4(up.Wp, up.wp_stride, 4 * i, wn);
    } else {
        const int unit = ((int)blockIdx.x - nbw) * 4 + (threadIdx.x >> 6);
        if (unit <= fin_units(f.fin)) finalize_unit(f.fin, unit, threadIdx.x & 63);
    }
}

int g_small_fin_lanes = 0;       // mdbn_set_option("small_fin_lanes"): 0 = by the number of partials

hipError_t launch_small_finish(const SmallFinArgs& f0, hipStream_t s)
{
    SmallFinArgs f = f0;
    f.lanes = f.nparts >= 64 ? 16 : f.nparts >= 24 ? 8 : f.nparts >= 12 ? 4 : f.nparts >= 4 ? 2 : 1;
    if (g_small_fin_lanes > 0) f.lanes = g_small_fin_lanes;
    const int ipb = 256 / f.lanes;
    const int nbw = (int)((f.n4p + ipb - 1) / ipb);
    const int nbf = ((int)((f.fin.ldh + f.fin.ldv + 15) / 16) + 1 + 3) / 4;       // fin_units + the cost unit, four waves per block
    hipLaunchKernelGGL(small_finish_kernel, dim3(nbw + nbf), dim3(256), 0, s, f);
    return hipGetLastError();
}

}  // namespace mdbn
