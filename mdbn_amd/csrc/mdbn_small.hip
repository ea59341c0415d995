// One launch per CD-k step for layers whose weights fit ONE CU's LDS (gfx950 / MI355X).
//
// The reference trains many such layers (MDBN.py:45-52, AMLsm2.py:242-340: 512 -> 40, 400 -> 40, 200 -> 20, 100 -> 24 -> 3, the
// joint layer), each step a scan of k gibbs_hvh (rbm.py:318-336).  On the multi-launch path such a step is 2 k + 4 dependent
// launches that each sit on the 5-7 us floor of a dependent tiny kernel (512 -> 40, CD-5, B = 512: 14 launches, 128 us for
// 0.27 GFLOP).  Here W (f32, <= ~100 KB) is staged into LDS once per workgroup, and a workgroup runs the WHOLE chain for
// 16-row slabs of the minibatch -- gather, positive phase, k x (propdown, propup) with the fused activations and the Philox
// draws, the reconstruction cost, and its share of the statistics S = v0^T ph - nv^T nh, accumulated in registers over its
// slabs -- on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) with every operand read from LDS.  It writes ONE partial
// [S | s_h | s_v | cost] per workgroup; a second, small launch (small_finish_kernel) sums the partials in workgroup order
// and applies the parameter update (or stores the statistics for a data-parallel all-reduce).  No workgroup waits for
// another.  Same Philox addressing as every other path (counter = (column, global row >> 2, draw, step)): the same uniforms
// meet probabilities that differ from the multi-launch path's by fp32 summation order only.
//
// MFMA operand layout (16x16x4 f32): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// D[m = 4 (lane >> 4) + e][n = lane & 15] -- a lane's four accumulator registers are four CONSECUTIVE ROWS of one column,
// exactly the four rows one Philox block serves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_small.h"

namespace mdbn {

typedef float sf32x4 __attribute__((ext_vector_type(4)));
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

#ifdef MDBN_STAMP       // diagnostic builds: shader-clock cycles workgroup 0 / wave 0 spends in the parts of a pass (slots 48..55)
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

// One pass of the chain, D[16][N] = A[16][K] * op(W), in two forms:
//
// sm_up (propup, K = V long, N = H a few tiles): work items = (group of TG tiles, K chunk) dealt over the waves -- the tiles of
// a group share the A operand (one LDS read feeds TG MFMAs) and give the wave TG independent accumulator chains; operands
// of round r + 1 are read before the MFMAs of round r are issued.  Partial tiles go through `part`; after a barrier the
// (row quad, column) pairs are dealt over the threads, each sums its ks partials in chunk order and applies `epi` ONCE
// (one copy of the epilogue's code: a workgroup runs every phase once per slab, its instructions are fetched cold).
//
// sm_down (propdown, K = H short, N = V many tiles): work items = groups of 4 tiles over the whole K; the wave applies `epi`
// to its accumulator registers (a lane holds rows 4 q .. 4 q + 3 of a column); `pre` is called before the reduction loop so
// that the epilogue's global loads fly under it.
//
// No operand masking anywhere: the pad columns of the 16-row buffers hold exact zeros (every writer keeps them so), W's
// index is clamped into the matrix, and a column n >= N computes a copy of column N - 1 that `epi` discards.
// The reduction loop of a pass: `rounds` rounds of two k-steps for TG tiles that share the A operand.  Two register sets
// alternate, so the operands of round r + 1 (and r + 2) are in flight while the MFMAs of round r issue, and no copy ties a
// round's MFMAs to the NEXT round's loads (hipcc's in-order lgkmcnt then waits only for the set it is about to use).  The
// look-ahead past the last round re-reads the last round (valid addresses, values unused) instead of branching.
template <int TG>
__device__ __forceinline__ void sm_kloop(sf32x4 (&acc)[TG], lds_cf* ap0, lds_cf* const (&bp0)[TG], int b1off, int bround, int rounds)
{
    struct Frag { float a0, a1, b0[TG], b1[TG]; };
    auto load = [&](Frag& f, int q) {
        f.a0 = ap0[8 * q]; f.a1 = ap0[8 * q + 4];
#pragma unroll
        for (int j = 0; j < TG; ++j) { f.b0[j] = bp0[j][q * bround]; f.b1[j] = bp0[j][q * bround + b1off]; }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < TG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0, f.b0[j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1, f.b1[j], acc[j], 0, 0, 0);
    };
    if (rounds <= 0) return;
    Frag f0, f1;
    load(f0, 0);
    for (int r = 0; r + 2 <= rounds; r += 2) {       // (scheduling barriers: hipcc otherwise sinks each load to its first use)
        load(f1, r + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, min(r + 2, rounds - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (rounds & 1) mma(f0);
}

template <int TG>
__device__ __forceinline__ void sm_up_loop(lds_cf* A, int lda, lds_cf* Wl, int ldw, int K, int N, int tiles, int ks,
                                           lds_f* part, int pld, int wave, int lane)
{
    const int c16 = lane & 15, kq = lane >> 4;
    const int ksteps = (K + 3) >> 2, kfull = K >> 2;
    const int per = (ksteps + ks - 1) / ks;
    const int groups = (tiles + TG - 1) / TG;
    const int bstep = 4 * ldw;
    for (int item = wave; item < groups * ks; item += SM_NW) {
        SM_CLK_BEGIN();
        const int grp = item % groups, ch = item / groups;
        const int s0 = ch * per, s1 = min(ksteps, s0 + per), s1f = min(s1, kfull);
        sf32x4 acc[TG];
        lds_cf* bp[TG];
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            acc[j] = sf32x4{0.f, 0.f, 0.f, 0.f};
            const int n = min((grp * TG + j) * 16 + c16, N - 1);
            bp[j] = Wl + (4 * s0 + kq) * ldw + n;
        }
        const int rounds = (s1f - s0) >> 1;
        sm_kloop<TG>(acc, A + c16 * lda + 4 * s0 + kq, bp, bstep, 2 * bstep, rounds);
        for (int s = s0 + 2 * rounds; s < s1; ++s) {     // odd step, and the K tail (W index clamped; A's pad is zero)
            const int k = 4 * s + kq, kc = k < K ? k : K - 1;
            const float av = A[c16 * lda + k];
#pragma unroll
            for (int j = 0; j < TG; ++j) {
                const int n = min((grp * TG + j) * 16 + c16, N - 1);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wl[kc * ldw + n], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int tile = grp * TG + j;
            if (tile < tiles) {
#pragma unroll
                for (int e = 0; e < 4; ++e) part[(ch * SM_ROWS + 4 * kq + e) * pld + tile * 16 + c16] = acc[j][e];
            }
        }
        SM_CLK_ADD(48);
    }
}

template <class Epi>
__device__ __forceinline__ void sm_up(lds_cf* A, int lda, lds_cf* Wl, int ldw, int K, int N, int tiles, int tg, int ks,
                                      lds_f* part, int pld, int wave, int lane, Epi&& epi)
{
    if (tg >= 4) sm_up_loop<4>(A, lda, Wl, ldw, K, N, tiles, ks, part, pld, wave, lane);
    else if (tg == 3) sm_up_loop<3>(A, lda, Wl, ldw, K, N, tiles, ks, part, pld, wave, lane);
    else if (tg == 2) sm_up_loop<2>(A, lda, Wl, ldw, K, N, tiles, ks, part, pld, wave, lane);
    else sm_up_loop<1>(A, lda, Wl, ldw, K, N, tiles, ks, part, pld, wave, lane);
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(49); }
    SM_CLK_BEGIN();
    const int ncol = tiles * 16;
    for (int q = threadIdx.x; q < 4 * ncol; q += SM_NT) {
        const int col = q % ncol, rq = q / ncol;
        sf32x4 x = {0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < ks; ++ch)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] += part[(ch * SM_ROWS + 4 * rq + e) * pld + col];
        epi(x, 4 * rq, col);
    }
    SM_CLK_ADD(50);
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(51); }
}

template <class Pre, class Epi>
__device__ __forceinline__ void sm_down(lds_cf* A, int lda, lds_cf* Wl, int ldw, int K, int N, int tiles,
                                        int wave, int lane, Pre&& pre, Epi&& epi)
{
    constexpr int TG = SM_TG;
    const int c16 = lane & 15, kq = lane >> 4;
    const int ksteps = (K + 3) >> 2, kfull = K >> 2;
    const int groups = (tiles + TG - 1) / TG;
    for (int grp = wave; grp < groups; grp += SM_NW) {
        SM_CLK_BEGIN();
        sf32x4 acc[TG], fetched[TG];
        lds_cf* bp[TG];
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            acc[j] = sf32x4{0.f, 0.f, 0.f, 0.f};
            const int n = (grp * TG + j) * 16 + c16;
            bp[j] = Wl + min(n, N - 1) * ldw + kq;
            fetched[j] = pre(4 * kq, n);
        }
        const int rounds = kfull >> 1;
        sm_kloop<TG>(acc, A + c16 * lda + kq, bp, 4, 8, rounds);
        for (int s = 2 * rounds; s < ksteps; ++s) {
            const int k = 4 * s + kq, kc = k < K ? k : K - 1;
            const float av = A[c16 * lda + k];
#pragma unroll
            for (int j = 0; j < TG; ++j) {
                const int n = min((grp * TG + j) * 16 + c16, N - 1);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Wl[n * ldw + kc], acc[j], 0, 0, 0);
            }
        }
        SM_CLK_ADD(52);
        const int nt = min(TG, tiles - grp * TG);
        for (int j = 0; j < nt; ++j) {                   // (a rolled loop: ONE copy of the epilogue's code)
            const sf32x4 x = j == 0 ? acc[0] : j == 1 ? acc[1] : j == 2 ? acc[2] : acc[3];
            const sf32x4 f = j == 0 ? fetched[0] : j == 1 ? fetched[1] : j == 2 ? fetched[2] : fetched[3];
            epi(x, f, 4 * kq, (grp * TG + j) * 16 + c16);
        }
        SM_CLK_ADD(53);
    }
    { SM_CLK_BEGIN(); SM_SYNC(); SM_CLK_ADD(54); }
}

// S tiles += X^T M over the slab's 16 rows.  Wave w owns the 16-row tiles ti = w, w + NW, ... of S (rt of them) times ALL TH
// tiles along H: the M fragments (TH x 4 k-steps) are read once per call, an X fragment serves TH MFMAs.
template <int TH>
__device__ __forceinline__ void sm_stats_t(lds_cf* Xl, int ldx, lds_cf* Ml, int ldh, int tiles_v,
                                           sf32x4 (&accS)[SM_MAXS], int wave, int lane)
{
    const int c16 = lane & 15, kq = lane >> 4;
    float b[TH][4];
#pragma unroll
    for (int tj = 0; tj < TH; ++tj)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[tj][s] = Ml[(4 * s + kq) * ldh + tj * 16 + c16];
#pragma unroll
    for (int r = 0; r < SM_MAXS / TH; ++r) {
        const int ti = wave + SM_NW * r;
        if (ti < tiles_v) {
            float av[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = Xl[(4 * s + kq) * ldx + ti * 16 + c16];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tj = 0; tj < TH; ++tj)
                    accS[r * TH + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b[tj][s], accS[r * TH + tj], 0, 0, 0);
        }
    }
}

// the finished S tiles of this wave into an LDS image [V][ld] (pad columns zero), for a coalesced copy to memory
template <int TH>
__device__ __forceinline__ void sm_park_t(lds_f* img, int ld, int V, int H, int tiles_v, const sf32x4 (&accS)[SM_MAXS], int wave, int lane)
{
    const int c16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int r = 0; r < SM_MAXS / TH; ++r) {
        const int ti = wave + SM_NW * r;
        if (ti < tiles_v) {
#pragma unroll
            for (int tj = 0; tj < TH; ++tj) {
                const int j = tj * 16 + c16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = ti * 16 + 4 * kq + e;
                    if (i < V && j < ld) img[i * ld + j] = j < H ? accS[r * TH + tj][e] : 0.f;
                }
            }
        }
    }
}

#define SM_TH_SWITCH(TH_, CALL)                                                               \
    switch (TH_) {                                                                            \
        case 1: { constexpr int TH = 1; CALL; } break;                                        \
        case 2: { constexpr int TH = 2; CALL; } break;                                        \
        case 3: { constexpr int TH = 3; CALL; } break;                                        \
        case 4: { constexpr int TH = 4; CALL; } break;                                        \
        case 5: { constexpr int TH = 5; CALL; } break;                                        \
        case 6: { constexpr int TH = 6; CALL; } break;                                        \
        case 7: { constexpr int TH = 7; CALL; } break;                                        \
        default: { constexpr int TH = 8; CALL; } break;                                       \
    }

// what the passes of one slab share (plain pointers and sizes: copied into registers once)
struct SmCtx {
    lds_f *Wl, *Xa, *Xb, *Hs, *Ml, *part, *csP, *csN, *csV, *hbl, *vbl;
    const int64_t* srcl;
    int row0; uint64_t grow0;
};

// v_t | h_{t-1}: RBM sigmoid + Bernoulli (draw 2t-1), GRBM linear mean (rbm.py:647-660, error_free).  LAST: the chain's last
// step also yields the reconstruction cost (rbm.py:372-374,449-482; GRBM :690-699: a sigmoid is applied to the linear
// mean), the column sums of v0 - nv, and (RBM) the visible MEAN for the statistics.  TAPS: inspection copies / chain taps.
// The flags are compile-time: a pass executes only the instructions it needs (at ~5 cycles per instruction and wave the
// generic epilogue cost as much as the pass's MFMAs).
template <bool GAUSS, bool LAST, bool TAPS>
__device__ __forceinline__ void sm_step_down(const SmallCdArgs& a, const SmCtx& c, int t, float& cost, int wave, int lane)
{
    const SmallLayout& L = a.L;
    const int V = a.V, B = a.B;
    const int64_t ldv = a.ldv;
    const sf32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    sm_down(c.Hs, L.ldhs, c.Wl, L.ldw, a.H, V, L.tiles_dn, wave, lane,
            [&](int r0, int col) -> sf32x4 {
                sf32x4 tg = zero4;          // the reconstruction target (v0 through the minibatch index), requested early
                if (LAST && col < V) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c.row0 + r0 + e < B) tg[e] = a.data[c.srcl[r0 + e] * a.ld_data + col];
                }
                return tg;
            },
            [&](const sf32x4& x, const sf32x4& tg, int r0, int col) {
                // (every LDS read first, every LDS write last: hipcc cannot tell the buffers apart and would otherwise
                // serialise a read behind each write -- a round trip per element)
                const bool live = col < V;
                const float bias = c.vbl[col];
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (!GAUSS) philox_rows4(a.rng, (uint32_t)(2 * t - 1), c.grow0 + (uint64_t)r0, (uint32_t)col, w);
                float m[4], sv[4], cs = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = live && c.row0 + r0 + e < B;
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
                    c.Xa[(r0 + e) * L.ldx + col] = GAUSS ? m[e] : sv[e];
                    if (!GAUSS && LAST) c.Xb[(r0 + e) * L.ldx + col] = m[e];
                }
                if (TAPS && col < (int)ldv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = c.row0 + r0 + e;
                        if (row < B) {
                            if (LAST && a.keep) a.V2[(int64_t)(B + row) * ldv + col] = m[e];
                            if (!GAUSS) {
                                if (a.keep) a.vs[(int64_t)row * ldv + col] = sv[e];
                                if (a.trace_v) a.trace_v[((int64_t)(t - 1) * B + row) * ldv + col] = sv[e];
                            }
                        }
                    }
                }
                if (LAST) {     // the four row quads of this column are lanes c, c + 16, c + 32, c + 48 of this wave
                    const float q0 = __shfl(cs, lane & 15, 64), q1 = __shfl(cs, (lane & 15) + 16, 64);
                    const float q2 = __shfl(cs, (lane & 15) + 32, 64), q3 = __shfl(cs, (lane & 15) + 48, 64);
                    if (lane < 16) c.csV[col] += (q0 + q1) + (q2 + q3);
                }
            });
}

// h_t | v_t (t = 0: from v0): from the mean for GRBM (rbm.py:669), from the sample for RBM (rbm.py:246).  KIND 0: the
// positive phase (mean kept for the statistics + sample, draw 0); 1: a middle step (sample only, draw 2t); 2: the chain's
// end (-nh for the statistics, no sample: CD does not materialise it).
template <int KIND, bool TAPS>
__device__ __forceinline__ void sm_step_up(const SmallCdArgs& a, const SmCtx& c, int t, int wave, int lane)
{
    const SmallLayout& L = a.L;
    const int H = a.H, B = a.B;
    const int64_t ldh = a.ldh;
    sm_up(c.Xa, L.ldx, c.Wl, L.ldw, a.V, H, L.tiles_up, L.tg_up, L.ks_up, c.part, L.pld, wave, lane,
          [&](const sf32x4& x, int r0, int col) {
              const bool live = col < H;
              const float bias = c.hbl[col];
              uint32_t w[4] = {0u, 0u, 0u, 0u};
              if (KIND != 2) philox_rows4(a.rng, (uint32_t)(2 * t), c.grow0 + (uint64_t)r0, (uint32_t)col, w);
              float m[4], sv[4], cs = 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                  const bool ok = live && c.row0 + r0 + e < B;
                  const float p = ok ? sigmoidf_(x[e] + bias) : 0.f;
                  m[e] = KIND == 2 ? -p : p;
                  cs += m[e];
                  sv[e] = (KIND != 2 && ok && philox_u01(w[e]) < p) ? 1.0f : 0.0f;
              }
              if (KIND != 1) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) c.Ml[(r0 + e) * L.ldhs + col] = m[e];
                  lds_f* cd = (KIND == 2 ? c.csN : c.csP) + (r0 >> 2) * L.Hp + col;
                  *cd += cs;
              }
              if (KIND != 2) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) c.Hs[(r0 + e) * L.ldhs + col] = sv[e];
              }
              if (TAPS && col < (int)ldh) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                      const int row = c.row0 + r0 + e;
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
__global__ __launch_bounds__(SM_NT) void small_cd_kernel(SmallCdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const SmallLayout& L = a.L;
    lds_f* const lds = (lds_f*)sm;
    SmCtx c;
    c.Wl = lds + L.oW;
    c.Xa = lds + L.oXa;             // visible operand of the next propup (v0; GRBM: nv mean; RBM: v sample)
    c.Xb = lds + L.oXb;             // RBM, last Gibbs step: nv MEAN (the statistics use the mean, the chain the sample)
    c.Hs = lds + L.oHs;             // hidden sample (operand of the next propdown)
    c.Ml = lds + L.oMl;             // hidden mean for the statistics: ph, then -nh
    c.part = lds + L.oPart;
    c.csP = lds + L.oCsP;           // [4][Hp] per-row-quad column sums of  ph
    c.csN = lds + L.oCsN;           // [4][Hp]                              -nh
    c.csV = lds + L.oCsV;           // [Vp] column sums of v0 - nv (the four row quads of a column sit in one wave)
    c.hbl = lds + L.oHb;
    c.vbl = lds + L.oVb;
    float* const red = sm + L.oRed;
    int64_t* const srcl = reinterpret_cast<int64_t*>(sm + L.oSrc);
    c.srcl = srcl;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int V = a.V, H = a.H, B = a.B;
    const int64_t ldv = a.ldv, ldh = a.ldh;
    const int nslabs = (B + SM_ROWS - 1) / SM_ROWS;
    const int k_steps = a.k;
    int n_stamp = 0;
    (void)n_stamp;
#ifdef MDBN_STAMP       // diagnostic builds (scripts/experiments/small_stamps.py): wall-clock stamps of workgroup 0's phases
    const long long sclk0 = clock64();
    if (blockIdx.x == 0 && tid == 0) { g_sm_clk = a.stamps; if (a.stamps) for (int i = 48; i < 56; ++i) a.stamps[i] = 0; }
#define SM_STAMP() do { if (a.stamps && blockIdx.x == 0 && tid == 0 && n_stamp < 46) a.stamps[1 + n_stamp++] = wall_clock64(); } while (0)
#else
#define SM_STAMP() do { } while (0)
#endif
    SM_STAMP();

    // ---- the first slab's source rows, then W (+ biases) and the slab's rows in ONE burst of loads; the column-sum partials
    //      start at zero.  W rows are float4-aligned in LDS (pitch ldw >= ldh; the pad columns of W are zero in memory)
    {   // W first (the long burst), the small loads behind it: vector-memory loads return in order, so a short dependent
        // load (index -> LDS) issued ahead of the burst would hold every W store back by its own latency
        const int q4 = (int)(ldh >> 2);
        for (int e = tid; e < V * q4; e += SM_NT) {
            const int r = e / q4, c4 = e - r * q4;
            *(lds_f4*)(c.Wl + r * L.ldw + 4 * c4) = *reinterpret_cast<const sf32x4*>(a.W + (int64_t)r * ldh + 4 * c4);
        }
    }
    if (tid < SM_ROWS) {
        const int row = (int)blockIdx.x * SM_ROWS + tid;
        srcl[tid] = row < B ? sm_src_row(a, row) : 0;
    }
    for (int e = tid; e < L.Hp; e += SM_NT) c.hbl[e] = e < H ? a.hbias[e] : 0.f;
    for (int e = tid; e < L.Vp; e += SM_NT) { c.vbl[e] = e < V ? a.vbias[e] : 0.f; c.csV[e] = 0.f; }
    for (int e = tid; e < 4 * L.Hp; e += SM_NT) { c.csP[e] = 0.f; c.csN[e] = 0.f; }
    sf32x4 accS[SM_MAXS];
#pragma unroll
    for (int u = 0; u < SM_MAXS; ++u) accS[u] = sf32x4{0.f, 0.f, 0.f, 0.f};
    float cost = 0.f;

    for (int slab = blockIdx.x; slab < nslabs; slab += gridDim.x) {
        c.row0 = slab * SM_ROWS;
        c.grow0 = a.rng.row_offset + (uint64_t)c.row0;                 // global row of the slab's first row (Philox address)
        if (slab != (int)blockIdx.x) {
            SM_SYNC();                                                // (the previous slab's last readers are done)
            if (tid < SM_ROWS) srcl[tid] = c.row0 + tid < B ? sm_src_row(a, c.row0 + tid) : 0;
        }
        SM_SYNC();
        SM_STAMP();
        // ---- x = train_set_x[indexes] (dbn.py:307): 16 rows into LDS
        {
            const int q4 = L.ldx >> 2, dq4 = (int)(ldv >> 2);
            for (int e = tid; e < SM_ROWS * q4; e += SM_NT) {
                const int r = e / q4, c4 = e - r * q4;
                sf32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c.row0 + r < B && c4 < dq4) {
                    v = *reinterpret_cast<const sf32x4*>(a.data + srcl[r] * a.ld_data + 4 * c4);
                    if (TAPS && a.keep) *reinterpret_cast<sf32x4*>(a.V2 + (int64_t)(c.row0 + r) * ldv + 4 * c4) = v;
                }
                *(lds_f4*)(c.Xa + r * L.ldx + 4 * c4) = v;
            }
        }
        SM_SYNC();
        SM_STAMP();
        // ---- positive phase (rbm.py:303), S += v0^T ph
        sm_step_up<0, TAPS>(a, c, 0, wave, lane);
        SM_STAMP();
        SM_TH_SWITCH(L.tiles_up, (sm_stats_t<TH>(c.Xa, L.ldx, c.Ml, L.ldhs, L.tiles_dn, accS, wave, lane)));
        SM_SYNC();
        SM_STAMP();
        // ---- k x gibbs_hvh (rbm.py:242-248, GRBM :662-671)
        for (int t = 1; t < k_steps; ++t) {
            sm_step_down<GAUSS, false, TAPS>(a, c, t, cost, wave, lane);
            SM_STAMP();
            sm_step_up<1, TAPS>(a, c, t, wave, lane);
            SM_STAMP();
        }
        sm_step_down<GAUSS, true, TAPS>(a, c, k_steps, cost, wave, lane);
        SM_STAMP();
        sm_step_up<2, TAPS>(a, c, k_steps, wave, lane);
        SM_STAMP();
        // ---- S += nv_mean^T (-nh_mean)
        SM_TH_SWITCH(L.tiles_up, (sm_stats_t<TH>(GAUSS ? c.Xa : c.Xb, L.ldx, c.Ml, L.ldhs, L.tiles_dn, accS, wave, lane)));
        SM_STAMP();
    }
    SM_SYNC();

    // ---- this workgroup's partials: S through an LDS image (the W region is free now) in whole 16-byte pieces, the column
    //      sums (quads combined in a fixed order), the cost
    {
        SM_TH_SWITCH(L.tiles_up, (sm_park_t<TH>(c.Wl, (int)ldh, V, H, L.tiles_dn, accS, wave, lane)));
        SM_SYNC();
        sf32x4* Sp = reinterpret_cast<sf32x4*>(a.part_S + (int64_t)blockIdx.x * V * ldh);
        const int n4 = V * (int)(ldh >> 2);
        for (int e = tid; e < n4; e += SM_NT) Sp[e] = *(const lds_f4*)(c.Wl + 4 * e);
        for (int j = tid; j < (int)ldh; j += SM_NT) {
            const bool in = j < L.Hp;
            a.posP[(int64_t)blockIdx.x * ldh + j] = in ? (c.csP[j] + c.csP[L.Hp + j]) + (c.csP[2 * L.Hp + j] + c.csP[3 * L.Hp + j]) : 0.f;
            a.negP[(int64_t)blockIdx.x * ldh + j] = in ? (c.csN[j] + c.csN[L.Hp + j]) + (c.csN[2 * L.Hp + j] + c.csN[3 * L.Hp + j]) : 0.f;
        }
        for (int i = tid; i < (int)ldv; i += SM_NT) a.partV[(int64_t)blockIdx.x * ldv + i] = i < L.Vp ? c.csV[i] : 0.f;
        const float tot = block_sum(cost, red);
        if (tid == 0) a.cost_partials[blockIdx.x] = tot;
    }
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
    if (B < 1 || V < 1 || H < 1 || V > 4096 || H > 4096) return false;
    const SmallLayout L = small_layout((int)V, (int)H, gauss != 0);
    if (L.bytes > SM_MAX_LDS) return false;
    return L.tiles_up <= SM_MAXTH && L.rt * L.tiles_up <= SM_MAXS;
}

hipError_t launch_small_cd(const SmallCdArgs& a, hipStream_t s)
{
    const SmallLayout L = small_layout(a.V, a.H, a.gauss != 0);
    if (!small_shape_ok(a.B, a.V, a.H, a.gauss)) return hipErrorInvalidValue;
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
    const dim3 grid(small_blocks(a.B)), block(SM_NT);
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
// Second launch: S = sum of the workgroups' partials (workgroup order), then either the parameter update of rbm.py:347-365
// on it (single device: update_rule4, as update_kernel) or a plain store into the statistics buffer (data-parallel: the
// all-reduce follows); the trailing blocks run the finalize units (bias statistics, cost, bias half of the update).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_finish_kernel(SmallFinArgs f)
{
    const int nbw = (int)((f.n4 + 255) / 256);
    if ((int)blockIdx.x < nbw) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= f.n4) return;
        const float4* P = reinterpret_cast<const float4*>(f.part);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = 0; p0 < f.nparts; p0 += 8) {         // 8 loads in flight, summed in workgroup order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = p0 + u < f.nparts ? P[i + (int64_t)(p0 + u) * (f.part_stride >> 2)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }
        }
        if (!f.do_upd) {
            reinterpret_cast<float4*>(f.S_out)[i] = s4;
            return;
        }
        const UpdEpi& u = f.upd;
        const float4 w = reinterpret_cast<const float4*>(u.W)[i], sp = reinterpret_cast<const float4*>(u.Ws)[i];
        const float4 w0 = u.W0 ? reinterpret_cast<const float4*>(u.W0)[i] : w;
        float4 wn, sn;
        update_rule4(w, sp, s4, w0, u.inv_bs, u.wc, upd_decay(u.lr, u.l2), u.l1, upd_two_lr_l1(u.lr, u.l1), u.mu, u.lr, wn, sn);
        reinterpret_cast<float4*>(u.W)[i] = wn;
        reinterpret_cast<float4*>(u.Ws)[i] = sn;
        if (u.Wp) store_planes4(u.Wp, u.wp_stride, 4 * i, wn);
    } else {
        const int unit = ((int)blockIdx.x - nbw) * 4 + (threadIdx.x >> 6);
        if (unit <= fin_units(f.fin)) finalize_unit(f.fin, unit, threadIdx.x & 63);
    }
}

hipError_t launch_small_finish(const SmallFinArgs& f, hipStream_t s)
{
    const int nbw = (int)((f.n4 + 255) / 256);
    const int nbf = ((int)((f.fin.ldh + f.fin.ldv + 15) / 16) + 1 + 3) / 4;       // fin_units + the cost unit, four waves per block
    hipLaunchKernelGGL(small_finish_kernel, dim3(nbw + nbf), dim3(256), 0, s, f);
    return hipGetLastError();
}

}  // namespace mdbn
