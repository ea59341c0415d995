// HIP kernels (gfx950 / MI355X) for the CD-k hot path of an RBM / GRBM.
//
// What each kernel replaces in the reference's Theano graph (src/rbm.py) is noted at
// its definition.  Layout facts used throughout:
//   * all matrices are float32 row-major with leading dimension ld (floats), ld % 4 == 0
//   * W is [V, ldh] (rbm.py:104); "K-contiguous" operands have the GEMM's reduction
//     index as the fastest dimension, "MN-contiguous" ones the output index
//   * the GEMM core is exact-f32 MFMA (v_mfma_f32_32x32x2_f32): 128x128 block tile,
//     4 waves (one per SIMD) each owning a 64x64 sub-tile = 2x2 MFMA accumulators,
//     BK = 32 slices staged through LDS (register prefetch, double-buffered LDS, one
//     barrier per slice), split-K so that ~one block per CU is in flight.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <stdint.h>
#include "philox.h"
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_skinny.h"

namespace mdbn {


constexpr int NTHREADS = 256;            // threads of one role (4 producer waves / 4 consumer waves)
// KB = slice depth along the reduction index (32 or 64); a K-contiguous LDS tile is [rows][KB + 1]
// (odd row stride: transposing 4-byte stores, conflict-free ds_read_b32 fragment reads)

// ----------------------------------------------------------------------------------
// operand staging: global -> registers -> LDS, done by the 256 threads of the producer
// waves.  ROWS = extent of the tile along the operand's output index (128 or 64); the
// reduction extent is the slice depth KB.  GUARD = false is the interior fast path (every
// float4 inside the operand: straight-line loads).
// ----------------------------------------------------------------------------------
template <int LAY, int ROWS, int KB, bool GUARD>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t ld, int MN, int K,
                                          int mn0, int k0, float4 (&r)[ROWS * KB / 1024])
{
    const int tid = threadIdx.x & (NTHREADS - 1);
    if constexpr (LAY == LAY_K) {
        // P[mn][k], k contiguous: KB/4 threads cover one KB-float row slice
        constexpr int TPRK = KB / 4, RPPK = NTHREADS / TPRK;
        const int c = tid % TPRK, rr = tid / TPRK;
        const int kq = k0 + 4 * c;
        const float* src = P + (int64_t)(mn0 + rr) * ld + kq;
#pragma unroll
        for (int p = 0; p < ROWS * KB / 1024; ++p) {
            const float* sp = src + (int64_t)(RPPK * p) * ld;
            if constexpr (!GUARD) {
                r[p] = *reinterpret_cast<const float4*>(sp);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mn0 + rr + RPPK * p < MN) {
                    if (kq + 3 < K) {
                        v = *reinterpret_cast<const float4*>(sp);
                    } else {
                        if (kq + 0 < K) v.x = sp[0];
                        if (kq + 1 < K) v.y = sp[1];
                        if (kq + 2 < K) v.z = sp[2];
                    }
                }
                r[p] = v;
            }
        }
    } else {
        // P[k][mn], mn contiguous: ROWS/4 threads cover one row of the tile
        constexpr int TPR = ROWS / 4, RPP = NTHREADS / TPR;
        const int c = tid % TPR, rr = tid / TPR;
        const int mq = mn0 + 4 * c;
        const float* src = P + (int64_t)(k0 + rr) * ld + mq;
#pragma unroll
        for (int p = 0; p < ROWS * KB / 1024; ++p) {
            const float* sp = src + (int64_t)(RPP * p) * ld;
            if constexpr (!GUARD) {
                r[p] = *reinterpret_cast<const float4*>(sp);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + rr + RPP * p < K) {
                    if (mq + 3 < MN) {
                        v = *reinterpret_cast<const float4*>(sp);
                    } else {
                        if (mq + 0 < MN) v.x = sp[0];
                        if (mq + 1 < MN) v.y = sp[1];
                        if (mq + 2 < MN) v.z = sp[2];
                    }
                }
                r[p] = v;
            }
        }
    }
}

template <int LAY, int ROWS, int KB>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&r)[ROWS * KB / 1024])
{
    const int tid = threadIdx.x & (NTHREADS - 1);
    if constexpr (LAY == LAY_K) {
        constexpr int TPRK = KB / 4, RPPK = NTHREADS / TPRK;
        const int c = tid % TPRK, rr = tid / TPRK;
#pragma unroll
        for (int p = 0; p < ROWS * KB / 1024; ++p) {
            float* d = T + (rr + RPPK * p) * (KB + 1) + 4 * c;     // odd stride: reads conflict-free
            d[0] = r[p].x; d[1] = r[p].y; d[2] = r[p].z; d[3] = r[p].w;
        }
    } else {
        constexpr int TPR = ROWS / 4, RPP = NTHREADS / TPR;
        const int c = tid % TPR, rr = tid / TPR;
#pragma unroll
        for (int p = 0; p < ROWS * KB / 1024; ++p)
        {
            const float4 v = make_float4(r[p].x, r[p].y, r[p].z, r[p].w);
            *reinterpret_cast<float4*>(T + (rr + RPP * p) * ROWS + 4 * c) = v;
        }
    }
}

// MFMA operand of lane (i = lane & 31, h = lane >> 5): element [mn = base + i][k = kk + h]
#ifndef MFMA_AGPR
#define MFMA_AGPR 0        // 1: inline-asm MFMA with the accumulators pinned to AGPRs
#endif
#ifndef CONSUMER_KG
#define CONSUMER_KG 2      // k-pairs per fragment group = prefetch distance of the LDS reads
#endif
#ifndef ABLATE_FRAG
#define ABLATE_FRAG 0    // diagnostic builds only: MFMA operands without LDS reads
#endif
template <int LAY, int ROWS, int KB>
__device__ __forceinline__ float frag(const float* __restrict__ T, int mn, int k)
{
#if ABLATE_FRAG == 1
    return (float)(mn + k) * 1e-3f;
#elif ABLATE_FRAG == 3
    return (float)(mn + k) * 1e-3f;
#elif ABLATE_FRAG == 2
    const float v = LAY == LAY_K ? T[mn * (KB + 1) + k] : T[k * ROWS + mn];
    asm volatile("" :: "v"(v));
    return (float)(mn + k) * 1e-3f;
#else
    return LAY == LAY_K ? T[mn * (KB + 1) + k] : T[k * ROWS + mn];
#endif
}

#ifdef MDBN_STAMP
// diagnostic build: s_memtime stamps of one block's phases (never compiled into the product)
#define STAMP(slot)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (g.stamps && blockIdx.x == 8 && (threadIdx.x & 63) == 0 && it < 64)               \
            g.stamps[(((threadIdx.x >> 6) * 64 + it) * 8) + (slot)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// ----------------------------------------------------------------------------------
// C[ks] = A[:, kchunk ks] * B[kchunk ks, :]   (split-K slabs, no epilogue)
// Replaces tensor.dot at rbm.py:168,198,226,411-412,650,685 and mlp.py:103.
//
// Wave-specialised block of 8 waves, one block per CU, block tile (64*MI) x (64*NI):
//   waves 0-3 (one per SIMD) CONSUME: fragment reads from LDS + v_mfma_f32_32x32x2_f32 only,
//             2x2 wave grid, MI x NI accumulators of 32x32 each; fragment reads run one
//             k-group ahead of the MFMAs so the matrix pipe never waits on LDS;
//   waves 4-7 PRODUCE: global -> registers -> LDS staging of the next KB-deep slice (its loads
//             issued a whole slice earlier), sharing each SIMD with one consumer wave whose
//             MFMA issue they barely disturb.
// Double-buffered LDS, one barrier per slice.  (Measured with s_memtime stamps: when every
// wave both staged and multiplied, a wave's ~100 staging instructions took ~3400 cycles per
// slice next to a partner wave's MFMAs and the matrix pipe idled 26-38% of the time.)
// ----------------------------------------------------------------------------------
constexpr int GEMM_THREADS = 512;

#ifndef ABLATE_STORE
#define ABLATE_STORE 0    // diagnostic builds only: timing ablations of the producer
#endif
#ifndef PRODUCER_SLEEP
#define PRODUCER_SLEEP 0    // x64 cycles of delay before the store burst (measured: any delay only hurts)
#endif
#ifndef PRODUCER_PRIO
#define PRODUCER_PRIO 3
#endif
#ifndef ABLATE_LOAD
#define ABLATE_LOAD 0
#endif

// PF (fused update, 128x128x64 interior tiles): the producers have nothing left to stage during the
// last slice and their 128 staging registers are free, so they fetch the tile's W / W_speed (16 + 16
// float4 per thread) while the consumers finish the MFMAs, and apply the update themselves once the
// tile is parked -- the read half of the read-modify-write is hidden behind the matrix pipe.
template <int LA, int LB, int MI, int NI, int KB, bool GUARD, bool PF = false>
__device__ __forceinline__ void gemm_produce(const GemmArgs& g, float* __restrict__ smem, int m0, int n0,
                                             int kbeg, int kend, int nt, bool pf = false)
{
    constexpr int BM = 64 * MI, BN = 64 * NI;
    constexpr int A_FLOATS = BM * (KB + 1), B_FLOATS = BN * (KB + 1), BUF = A_FLOATS + B_FLOATS;
    // Two register sets: in slice `it` the set (it & 1) holds slice it+1 (loaded two slices
    // ago), is written to LDS buffer (it+1) & 1, and is then re-loaded with slice it+3 --
    // every load has more than a full slice (~2 us) to land before it is needed.
    float4 ra0[BM * KB / 1024], rb0[BN * KB / 1024], ra1[BM * KB / 1024], rb1[BN * KB / 1024];
#define PRODUCER_LOAD(RA, RB, SLICE)                                                          \
    do {                                                                                      \
        load_tile<LA, BM, KB, GUARD>(g.A, g.lda, g.M, kend, m0, kbeg + (SLICE) * KB, RA);         \
        load_tile<LB, BN, KB, GUARD>(g.B, g.ldb, g.N, kend, n0, kbeg + (SLICE) * KB, RB);         \
    } while (0)
#define PRODUCER_STEP(RA, RB)                                                                 \
    do {                                                                                      \
        STAMP(0);                                                                             \
        __builtin_amdgcn_s_sleep(PRODUCER_SLEEP);                                             \
        if (it + 1 < nt && !ABLATE_STORE) {                                                   \
            float* nx = smem + ((it + 1) & 1) * BUF;   /* all reads of it ended at the last barrier */ \
            store_tile<LA, BM, KB>(nx, RA);                                                       \
            store_tile<LB, BN, KB>(nx + A_FLOATS, RB);                                            \
        }                                                                                     \
        STAMP(1);                                                                             \
        if (it + 3 < nt && !ABLATE_LOAD) PRODUCER_LOAD(RA, RB, it + 3);                       \
        STAMP(2);                                                                             \
        __syncthreads();                                                                      \
        STAMP(3);                                                                             \
    } while (0)

    PRODUCER_LOAD(ra0, rb0, 0);
    store_tile<LA, BM, KB>(smem, ra0);
    store_tile<LB, BN, KB>(smem + A_FLOATS, rb0);
    if (nt > 1) PRODUCER_LOAD(ra0, rb0, 1);         // slices 1 and 2 stay in flight across the barrier
    if (nt > 2) PRODUCER_LOAD(ra1, rb1, 2);
    __syncthreads();
    const int nloop = (PF && pf) ? nt - 1 : nt;     // PF: the last step (no store, no load) is peeled below
    for (int it = 0; it < nloop; ++it) {
        PRODUCER_STEP(ra0, rb0);
        if (++it >= nloop) break;
        PRODUCER_STEP(ra1, rb1);
    }
#undef PRODUCER_STEP
#undef PRODUCER_LOAD
    if constexpr (PF) {
        if (pf) {
            static_assert(!PF || (BM == 128 && BN == 128), "PF is written for the 128x128 tile");
            constexpr int LDT = BN + 8, ROWS_PER_PASS = NTHREADS / (BN / 4), NP = BM / ROWS_PER_PASS;
            const UpdEpi& u = g.upd;
            const int tid = threadIdx.x & (NTHREADS - 1);
            const int c4 = tid % (BN / 4), rr = tid / (BN / 4);
            const int64_t base = (int64_t)(m0 + rr) * u.ld + n0 + 4 * c4;
            float4 w[NP], sp[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t off = base + (int64_t)(ROWS_PER_PASS * p) * u.ld;
                w[p] = *reinterpret_cast<const float4*>(u.W + off);
                sp[p] = *reinterpret_cast<const float4*>(u.Ws + off);
            }
            __syncthreads();                // end of the last slice
            __syncthreads();                // the consumers parked the tile
            const float two_lr_l1 = upd_two_lr_l1(u.lr, u.l1);
            const float decay = upd_decay(u.lr, u.l2);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t off = base + (int64_t)(ROWS_PER_PASS * p) * u.ld;
                const float4 st = *reinterpret_cast<const float4*>(smem + (rr + ROWS_PER_PASS * p) * LDT + 4 * c4);
                float4 wn, sn;
                update_rule4(w[p], sp[p], st, w[p], u.inv_bs, u.wc, decay, u.l1, two_lr_l1, u.mu, u.lr, wn, sn);
                *reinterpret_cast<float4*>(u.W + off) = wn;
                *reinterpret_cast<float4*>(u.Ws + off) = sn;
                if (u.Wp) store_planes4(u.Wp, u.wp_stride, off, wn);
            }
        }
    }
}

// BMI / BNI: block tile in 64-row / 64-column units (LDS image); MI / NI: this wave's 32x32 accumulators
template <int LA, int LB, int BMI, int BNI, int MI, int NI, int KB>
__device__ __forceinline__ void gemm_consume(const GemmArgs& g, const float* __restrict__ smem,
                                             f32x16 (&acc)[MI][NI], int nt, int wm, int wn, int i, int h)
{
    constexpr int BM = 64 * BMI, BN = 64 * BNI;
    constexpr int A_FLOATS = BM * (KB + 1), B_FLOATS = BN * (KB + 1), BUF = A_FLOATS + B_FLOATS;
    constexpr int KG = CONSUMER_KG, NG = KB / (2 * KG);   // groups of KG k-pairs per slice
    __syncthreads();                                // slice 0 staged
    for (int it = 0; it < nt; ++it) {
        STAMP(0);
        const float* at = smem + (it & 1) * BUF;
        const float* bt = at + A_FLOATS;
        float av[2][KG][MI], bv[2][KG][NI];
#pragma unroll
        for (int u = 0; u < KG; ++u) {
#pragma unroll
            for (int a = 0; a < MI; ++a) av[0][u][a] = frag<LA, BM, KB>(at, wm + 32 * a + i, 2 * u + h);
#pragma unroll
            for (int b = 0; b < NI; ++b) bv[0][u][b] = frag<LB, BN, KB>(bt, wn + 32 * b + i, 2 * u + h);
        }
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            const int cs = grp & 1, ns = cs ^ 1;
            if (grp + 1 < NG) {
#pragma unroll
                for (int u = 0; u < KG; ++u) {
                    const int kk = 2 * KG * (grp + 1) + 2 * u;
#pragma unroll
                    for (int a = 0; a < MI; ++a) av[ns][u][a] = frag<LA, BM, KB>(at, wm + 32 * a + i, kk + h);
#pragma unroll
                    for (int b = 0; b < NI; ++b) bv[ns][u][b] = frag<LB, BN, KB>(bt, wn + 32 * b + i, kk + h);
                }
            }
#pragma unroll
            for (int u = 0; u < KG; ++u)
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int b = 0; b < NI; ++b)
#if MFMA_AGPR
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
                                     : "+a"(acc[a][b]) : "v"(av[cs][u][a]), "v"(bv[cs][u][b]));
#else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][u][a], bv[cs][u][b], acc[a][b], 0, 0, 0);
#endif
#if ABLATE_FRAG == 3
            {   // same bytes as the real fragment reads, fetched as 16-byte reads (dummy data)
                const int ln = threadIdx.x & 63;
#pragma unroll
                for (int q = 0; q < KG * (MI + NI) / 4; ++q) {
                    const float4 d = *reinterpret_cast<const float4*>(at + 4 * ln + 256 * (grp * 2 + q));
                    asm volatile("" :: "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
                }
            }
#endif
            // emitted order: one fragment read of the NEXT group behind every MFMA of this group,
            // so each read issues in the 64-cycle shadow of a running MFMA (issued as a block
            // after the MFMAs, the reads cost the matrix pipe ~18% -- diagnostic ablation)
#pragma unroll
            for (int m = 0; m < KG * MI * NI; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        STAMP(1);
        __syncthreads();
        STAMP(2);
    }
}

// CW: consumer (MFMA) waves per SIMD.  1: four consumers with a 2x2 grid of (32*MI)x(32*NI) wave tiles.
// 2: eight consumers, 2x4 grid of (32*MI)x(16*NI) wave tiles -- the two MFMA waves of a SIMD cover each
// other's LDS-return stalls (needs NI == 2; unfused kernels only).
template <int LA, int LB, int MI, int NI, int KB, int FUSED, int CW>
__global__ __launch_bounds__(64 * (4 * CW + 4)) void gemm_splitk_kernel(GemmArgs g)
{
    MDBN_GEMM_ARGS_EARLY(g);
    constexpr int BM = 64 * MI, BN = 64 * NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // XCD-aware work mapping: blocks b and b+8 share an XCD (round-robin dispatch), so
    // give each XCD one contiguous chunk of the (ks, tile) list -> its blocks share A/B
    // panels through that XCD's L2.  Placement only affects speed, never results.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.inner_m) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else           { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = ks * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg + KB - 1) / KB;     // >= 1: the host never plans an empty split

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WMI = MI, WNI = CW == 2 ? NI / 2 : NI;       // this wave's accumulator grid
    static_assert(CW == 1 || (CW == 2 && NI == 2 && FUSED == 0), "CW == 2 needs a 128-column unfused tile");
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N) && (kbeg + nt * KB <= g.K);
    // producer-prefetched update (see gemm_produce): full 128x128x64 tiles, live weight cost only
    constexpr bool PF = FUSED == 2 && KB == 64 && MI == 2 && NI == 2;
    const bool pf = PF && interior && g.upd.W0 == nullptr;
    if (wave >= 4 * CW) {
        // the few staging instructions must not queue behind the partner wave's MFMA stream
        __builtin_amdgcn_s_setprio(PRODUCER_PRIO);
        if (interior) gemm_produce<LA, LB, MI, NI, KB, false, PF>(g, smem, m0, n0, kbeg, kend, nt, pf);
        else gemm_produce<LA, LB, MI, NI, KB, true>(g, smem, m0, n0, kbeg, kend, nt);
        if constexpr (FUSED == 0) return;
        if (pf) return;                     // this thread already applied its share of the update
        __builtin_amdgcn_s_setprio(0);
    } else {
        const int lane = threadIdx.x & 63;
        const int i = lane & 31, h = lane >> 5;
        const int wm = CW == 2 ? (wave >> 2) * (32 * WMI) : (wave >> 1) * (32 * WMI);
        const int wn = CW == 2 ? (wave & 3) * (32 * WNI) : (wave & 1) * (32 * WNI);
        f32x16 acc[WMI][WNI];
#pragma unroll
        for (int a = 0; a < WMI; ++a)
#pragma unroll
            for (int b = 0; b < WNI; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

        if constexpr (FUSED == 2 || (FUSED == 0 && LA == LAY_MN)) {
            // statistics GEMM: bias statistics, cost (and, fused step, the bias update) ride on the
            // consumers' idle ramp-up (the first slice is still on its way from HBM): wave-sized units
            // spread over all blocks, wave 0 of every block first
            if (g.fin_enabled) {
                const int nu = fin_units(g.fin);
                for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += 4 * CW * (int)gridDim.x)
                    finalize_unit(g.fin, unit, lane);
            }
        }
        gemm_consume<LA, LB, MI, NI, WMI, WNI, KB>(g, smem, acc, nt, wm, wn, i, h);
#if MFMA_AGPR
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA result -> v_accvgpr_read hazard (asm is opaque to hipcc)
#endif

        // accumulator (32x32): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        if constexpr (FUSED != 0) {
            // Park the tile in LDS (every read of the last slice ended at the loop's final barrier).
            // Row stride BN + 8: lanes 32..63 (rows + 4) land 32 banks away from lanes 0..31.
            constexpr int LDT = BN + 8;
#pragma unroll
            for (int a = 0; a < WMI; ++a)
#pragma unroll
                for (int b = 0; b < WNI; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        smem[(wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h) * LDT + wn + 32 * b + i] = acc[a][b][e];
        } else {
            float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
            for (int a = 0; a < WMI; ++a)
#pragma unroll
                for (int b = 0; b < WNI; ++b) {
                    const int col = n0 + wn + 32 * b + i;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (row < g.M && col < g.Nst) C[(int64_t)row * g.ldc + col] = acc[a][b][e];
                    }
                }
            return;
        }
    }
    if constexpr (FUSED != 0) {     // all 8 waves work on the parked tile
        __syncthreads();
        if (pf) return;             // (consumers) the producers hold W / W_speed in registers and finish
        if constexpr (FUSED == 1) fused_tile_epilogue<BM, BN>(g.epi, smem, m0, n0);   // activation + sampling
        else fused_update_epilogue<BM, BN>(g.upd, smem, m0, n0);                       // parameter update
    }
}

template <int LA, int LB, int MI, int NI, int KB, int FUSED, int CW = 1>
static hipError_t launch_gemm_t(const GemmArgs& g, hipStream_t s)
{
    // The LDS request is padded past half of the CU's 160 KiB so that exactly one 8-wave
    // block lives on a CU: the jobs then spread evenly over the 256 CUs.
    constexpr int tile_bytes = 2 * (64 * MI + 64 * NI) * (KB + 1) * (int)sizeof(float);
    constexpr int park_bytes = FUSED != 0 ? (64 * MI * (64 * NI + 8) + 8) * (int)sizeof(float) : 0;
    constexpr int need_bytes = tile_bytes > park_bytes ? tile_bytes : park_bytes;
    // FUSED == 2 with 32-deep slices is the statistics GEMM of a small batch (K = 2B < 128): a few
    // slices, then a long read-modify-write epilogue -- let two blocks share a CU so one block's
    // epilogue overlaps the other's loads
    constexpr bool two_per_cu = FUSED == 2 && KB == 32;
    constexpr int lds_bytes = two_per_cu ? need_bytes : (need_bytes > 84 * 1024 ? need_bytes : 84 * 1024);
    static bool attr_set = false;
    auto kern = gemm_splitk_kernel<LA, LB, MI, NI, KB, FUSED, CW>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = g.tiles_m * g.tiles_n * g.splitk;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (4 * CW + 4)), lds_bytes, s, g);
    return hipGetLastError();
}

template <int LA, int LB, int FUSED>
static hipError_t launch_gemm_l(const GemmArgs& g, hipStream_t s)
{
    if constexpr (FUSED == 0) {
        if (g.cw == 2 && g.bn == 128 && g.bk == 64) return launch_gemm_t<LA, LB, 2, 2, 64, 0, 2>(g, s);
        if (g.cw == 2 && g.bn == 128 && g.bk == 32) return launch_gemm_t<LA, LB, 2, 2, 32, 0, 2>(g, s);
    }
    if (g.bn == 128 && g.bk == 64) return launch_gemm_t<LA, LB, 2, 2, 64, FUSED>(g, s);
    if (g.bn == 128 && g.bk == 32) return launch_gemm_t<LA, LB, 2, 2, 32, FUSED>(g, s);
    if (g.bn == 64 && g.bk == 32) return launch_gemm_t<LA, LB, 2, 1, 32, FUSED>(g, s);
    return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------------
// GEMM on the bf16 matrix pipe at f32 accuracy ("bf16x6"): the tiled kernel's operands (f32 in HBM,
// K-contiguous or row-contiguous) and outputs (split-K slabs, plain C, or the fused statistics epilogues).
// Every f32 value is split into three bf16 pieces x = x1 + x2 + x3 (exact: bf16 keeps f32's exponent)
// and the six piece products with i + j <= 4 run on v_mfma_f32_32x32x16_bf16 with f32 accumulation;
// the dropped products are <= 3 * 2^-24 relative, the order of f32 rounding itself (measured against
// float64: the same error as the exact-f32 kernel and rocBLAS, scripts/experiments/bf16x6_gemm.py).
// Six 32-cycle MFMAs per 16 k replace eight 64-cycle ones.
//
// The split happens in the producer waves, together with the transposition the MFMA operand layout
// needs (8 consecutive k per lane = one 16-byte LDS read): a producer thread loads 8 k-rows x 4
// columns (8 coalesced float4), splits them with v_cvt_pk_bf16_f32 (4.5 VALU ops per element),
// transposes in registers and writes, per column and piece, ONE ds_write_b128 of 8 k-values.  LDS
// plane tile: [128 rows][32 k] bf16, 80-byte row pitch; the 16-byte k-chunk index is XOR-swizzled
// with (row >> 4) & 3 so that both the transposing writes (rows 4 apart per lane) and the fragment
// reads (rows 1 apart) are bank-conflict free.  Nothing upstream changes: operands stay f32 in HBM.
// 128x128 block tile, 32-deep slices, 4 producer + 4 consumer waves (2x2 grid of 64x64 wave tiles),
// double-buffered LDS; epilogues as the tiled kernel (split-K slab / plain store, FUSED = 1: activation
// + sampling on the parked tile of an unsplit forward pass, FUSED = 2: finalize
// units in the consumers' ramp-up and the parameter update on the parked tile).  Edge tiles and the K
// tail take a guarded producer (rows / k beyond the operand load as zero).
// ----------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4n __attribute__((ext_vector_type(4)));

constexpr int X6_KB = 32, X6_ROWB = 80, X6_PLANE = 128 * X6_ROWB, X6_BUF = 6 * X6_PLANE;

// split two floats into three packed-bf16 pairs (lo half = first value)
__device__ __forceinline__ void x6_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3)
{
    // Truncating split on full-rate integer / f32 instructions: piece = value & 0xffff0000 (the upper half of
    // an f32 IS a bf16), remainder = value - piece (exact).  Three pieces of 8 significant bits cover the
    // 24-bit significand exactly; v_perm_b32 packs the upper halves of two values into one dword.  (The
    // rounding conversion v_cvt_pk_bf16_f32 and v_dot2c_f32_bf16 are quarter-rate: with them the producer
    // waves needed ~2300 cycles per slice for split + store and the MFMA waves waited at every barrier.)
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u);
    const float rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
    const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u);
    const float sb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);            // (hi16(b) << 16) | hi16(a)
    p2 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int NDW>
__device__ __forceinline__ void x6_store_piece(unsigned char* d, const unsigned (&p)[NDW])
{
    if constexpr (NDW == 4) { const u32x4 v = {p[0], p[1], p[2], p[3]}; *reinterpret_cast<u32x4*>(d) = v; }
    else { const u32x2 v = {p[0], p[1]}; *reinterpret_cast<u32x2*>(d) = v; }
}

// One operand's staging loop, run by PW producer waves (64 * PW threads) per operand.  LX = LAY_MN:
// P[k][ld], rows contiguous -- a thread owns rows 4*mg..+3 and 16/PW consecutive k-rows, loads them as
// float4 and transposes in registers.  LX = LAY_K: P[row][ld], k contiguous -- a thread owns one k-octet of
// 8/PW rows and loads two float4 per row.  16/PW float4 per thread and slice either way.
// NP = pieces stored (3, or 1 for an operand whose values are exactly representable in bf16: the 0/1
// samples of a Gibbs chain -- the other two pieces would be zero)
// GUARD: rows >= MN and k >= kend load as zero (edge tiles / K tail); false = interior fast path
template <int LX, int NP = 3, bool GUARD = false, int PW = 2>
__device__ __forceinline__ void x6_produce(const float* __restrict__ P, int64_t ld, int row0, int kbeg, int nt,
                                           unsigned char* __restrict__ planes /* this operand's 3 planes, buffer 0 */,
                                           int t /* thread within the operand's producers */,
                                           int MN = 0, int kend = 0, unsigned long long* stamp_buf = nullptr)
{
    static_assert(PW == 2 || PW == 4, "2 or 4 producer waves per operand");
    const f32x4n zero4 = {0.f, 0.f, 0.f, 0.f};
    const struct { unsigned long long* stamps; } g = {stamp_buf};      // for STAMP() (diagnostic builds)
    (void)g;
    constexpr int KB = X6_KB, NF = 16 / PW;         // float4 per thread and slice
    f32x4n r0[NF], r1[NF];                          // two register sets: two slices ahead
#ifdef MDBN_STAMP
#define X6_DIAG_WAIT() do { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); STAMP(4); } while (0)
#else
#define X6_DIAG_WAIT() do { } while (0)
#endif
#define X6_PIPELINE()                                                                             \
    X6_LOAD(r0, 0);                                                                               \
    X6_STORE(r0, 0);                                                                              \
    if (nt > 1) { X6_LOAD(r0, 1); }                                                               \
    if (nt > 2) { X6_LOAD(r1, 2); }                                                               \
    __syncthreads();                                                                              \
    for (int it = 0; it < nt; ++it) {                                                             \
        STAMP(0);                                                                                 \
        X6_DIAG_WAIT();                                                                           \
        if (it + 1 < nt) { X6_STORE(r0, (it + 1) & 1); }                                          \
        STAMP(1);                                                                                 \
        if (it + 3 < nt) { X6_LOAD(r0, it + 3); }                                                 \
        STAMP(2);                                                                                 \
        __syncthreads();                                                                          \
        STAMP(3);                                                                                 \
        if (++it >= nt) break;                                                                    \
        STAMP(0);                                                                                 \
        X6_DIAG_WAIT();                                                                           \
        if (it + 1 < nt) { X6_STORE(r1, (it + 1) & 1); }                                          \
        STAMP(1);                                                                                 \
        if (it + 3 < nt) { X6_LOAD(r1, it + 3); }                                                 \
        STAMP(2);                                                                                 \
        __syncthreads();                                                                          \
        STAMP(3);                                                                                 \
    }
    if constexpr (LX == LAY_MN) {
        // NKR k-rows per thread: a whole k-octet (one 16-byte LDS write per piece) or half of one (8 bytes)
        constexpr int NKR = NF, NDW = NKR / 2;
        const int mg = t & 31, kg = t >> 5;                     // column group, k-row group
        const int chunk = (kg * NKR) >> 3, half = (kg * NKR) & 7;    // 16-byte chunk of the slice, k offset inside
        const float* src = P + row0 + 4 * mg + (int64_t)(kbeg + kg * NKR) * ld;
        unsigned char* const dst0 = planes + (4 * mg) * X6_ROWB + ((chunk ^ ((mg >> 2) & 3)) << 4) + 2 * half;
        const bool col_ok = !GUARD || row0 + 4 * mg < MN;      /* pad columns up to ld are zero in memory */
#define X6_LOAD(R, SLICE)                                                                         \
    _Pragma("unroll") for (int rr = 0; rr < NKR; ++rr) {                                          \
        const bool ok = !GUARD || (col_ok && kbeg + kg * NKR + (SLICE) * KB + rr < kend);         \
        R[rr] = ok ? *reinterpret_cast<const f32x4n*>(src + (int64_t)((SLICE) * KB + rr) * ld) : zero4; \
    }
#define X6_STORE(R, BUF)                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        unsigned p1[NDW], p2[NDW], p3[NDW];                                                       \
        _Pragma("unroll") for (int kk = 0; kk < NDW; ++kk)                                        \
            x6_split2(R[2 * kk][j], R[2 * kk + 1][j], p1[kk], p2[kk], p3[kk]);                    \
        unsigned char* d = dst0 + (BUF) * X6_BUF + j * X6_ROWB;                                   \
        x6_store_piece<NDW>(d, p1);                                                               \
        if (NP == 3) {                                                                            \
            x6_store_piece<NDW>(d + X6_PLANE, p2);                                                \
            x6_store_piece<NDW>(d + 2 * X6_PLANE, p3);                                            \
        }                                                                                         \
    }
        X6_PIPELINE()
#undef X6_LOAD
#undef X6_STORE
    } else {
        constexpr int NRW = NF / 2, RSTRIDE = 16 * PW;          // rows per thread, distance between them
        const int ko = t & 3, rb = t >> 2;
        const float* src = P + (int64_t)(row0 + rb) * ld + kbeg + ko * 8;
        unsigned char* const dst0 = planes + rb * X6_ROWB;
#define X6_LOAD(R, SLICE)                                                                         \
    _Pragma("unroll") for (int jj = 0; jj < NRW; ++jj) {                                          \
        const float* sp = src + (int64_t)(RSTRIDE * jj) * ld + (SLICE) * KB;                      \
        const int k0 = kbeg + ko * 8 + (SLICE) * KB;                                              \
        const bool row_ok = !GUARD || row0 + rb + RSTRIDE * jj < MN;                              \
        R[2 * jj] = (row_ok && (!GUARD || k0 < kend)) ? *reinterpret_cast<const f32x4n*>(sp) : zero4;        \
        R[2 * jj + 1] = (row_ok && (!GUARD || k0 + 4 < kend)) ? *reinterpret_cast<const f32x4n*>(sp + 4) : zero4; \
    }
#define X6_STORE(R, BUF)                                                                          \
    _Pragma("unroll") for (int jj = 0; jj < NRW; ++jj) {                                          \
        unsigned p1[4], p2[4], p3[4];                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                          \
            x6_split2(R[2 * jj + (kk >> 1)][2 * (kk & 1)], R[2 * jj + (kk >> 1)][2 * (kk & 1) + 1], p1[kk], p2[kk], p3[kk]); \
        const int row = rb + RSTRIDE * jj;                                                        \
        unsigned char* d = dst0 + (BUF) * X6_BUF + (RSTRIDE * jj) * X6_ROWB + ((ko ^ ((row >> 4) & 3)) << 4); \
        x6_store_piece<4>(d, p1);                                                                 \
        if (NP == 3) {                                                                            \
            x6_store_piece<4>(d + X6_PLANE, p2);                                                  \
            x6_store_piece<4>(d + 2 * X6_PLANE, p3);                                              \
        }                                                                                         \
    }
        X6_PIPELINE()
#undef X6_LOAD
#undef X6_STORE
    }
#undef X6_PIPELINE
#undef X6_DIAG_WAIT
}

// AP = pieces of the A operand (3, or 1 when A holds 0/1 samples: three products instead of six).
// RAGGED = false: the host guarantees whole tiles and slices (no guarded code in the kernel at all: the
// guarded variant costs the whole-tile case 2.6% through register allocation).
// PW = producer waves per operand (2: 8-wave block, 4: 12-wave block -- the producers' per-slice latency,
// which the MFMA waves wait for at the barrier, halves).
#ifndef X6_STAGE_ACC
#define X6_STAGE_ACC 1
#endif
template <int LA, int LB, int FUSED, int AP = 3, bool RAGGED = false, int PW = 2>
__global__ __launch_bounds__(64 * (4 + 2 * PW)) void gemm_bf16x6_kernel(GemmArgs g)
{
    MDBN_GEMM_ARGS_EARLY(g);
    constexpr int BM = 128, BN = 128, KB = X6_KB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);

    // XCD-aware (split, tile) order, as gemm_splitk_kernel
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qq = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.inner_m) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else           { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = ks * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nt = (kend - kbeg + KB - 1) / KB;                  // K tail: zero-filled by the guarded producer
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool interior = !RAGGED || ((m0 + BM <= g.M) && (n0 + BN <= g.N) && (kbeg + nt * KB <= g.K));

    if (wave >= 4) {
        __builtin_amdgcn_s_setprio(PRODUCER_PRIO);
        const bool opA = wave < 4 + PW;
        const int t = (int)threadIdx.x - 256 - (opA ? 0 : 64 * PW);
        if (!RAGGED || interior) {
            if (opA) x6_produce<LA, AP, false, PW>(g.A, g.lda, m0, kbeg, nt, lds, t, 0, 0, g.stamps);
            else x6_produce<LB, 3, false, PW>(g.B, g.ldb, n0, kbeg, nt, lds + 3 * X6_PLANE, t, 0, 0, g.stamps);
        } else if constexpr (RAGGED) {
            if (opA) x6_produce<LA, AP, true, PW>(g.A, g.lda, m0, kbeg, nt, lds, t, g.M, kend);
            else x6_produce<LB, 3, true, PW>(g.B, g.ldb, n0, kbeg, nt, lds + 3 * X6_PLANE, t, g.N, kend);
        }
        if constexpr (FUSED == 0) return;
        __builtin_amdgcn_s_setprio(0);
    } else {
        // ---- consumers
        const int lane = threadIdx.x & 63, i = lane & 31, q = lane >> 5;
        const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
        if constexpr (LA == LAY_MN) {
            if (g.fin_enabled) {        // statistics GEMM: finalize units on the idle ramp-up
                const int nu = fin_units(g.fin);
                for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += 4 * (int)gridDim.x)
                    finalize_unit(g.fin, unit, lane);
            }
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        // per-lane fragment addresses: row pitch 80 B, 16-byte chunk (2 s + q) ^ ((row >> 4) & 3)
        int offA[2][2], offB[2][2];                 // [32-row block][k-step s]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                const int ra = wm + 32 * a + i, rb = wn + 32 * a + i;
                offA[a][sidx] = ra * X6_ROWB + (((2 * sidx + q) ^ ((ra >> 4) & 3)) << 4);
                offB[a][sidx] = 3 * X6_PLANE + rb * X6_ROWB + (((2 * sidx + q) ^ ((rb >> 4) & 3)) << 4);
            }
        // Software pipeline over the two 16-k steps of a slice, with the slice barrier BETWEEN them:
        //   F1 <- fragments(step 1, slice it)         issued under the MFMAs of step 0
        //   MFMAs(F0)
        //   barrier: every consumer holds all of slice it in registers (its LDS buffer may be refilled),
        //            the producers have finished slice it + 1
        //   F0 <- fragments(step 0, slice it + 1)     issued under the MFMAs of step 1
        //   MFMAs(F1)
        // The F0 reload is unconditional: guarded by `it + 1 < nt` the kernel produced garbage (fragment
        // registers reloaded in a separate basic block while MFMAs of the previous block still read them);
        // on the last slice it reads the other (stale) buffer and the values are never used.
        bf16x8 f0a[3][2], f0b[3][2], f1a[3][2], f1b[3][2];
#define X6_FRAGS(FA, FB, BASE, SIDX)                                                              \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                              \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                           \
            if (pl < AP) FA[pl][a] = *reinterpret_cast<const bf16x8*>((BASE) + pl * X6_PLANE + offA[a][SIDX]); \
            FB[pl][a] = *reinterpret_cast<const bf16x8*>((BASE) + pl * X6_PLANE + offB[a][SIDX]);  \
        }
        // STAGE_ACC (propup layout: the long reductions over the visible units): the products of one 16-k step are chained
        // into a step-local accumulator that starts at zero and the step's sum is added to the tile's accumulator by the
        // VALU -- the tile's accumulator takes one rounding at its own magnitude per step instead of six (three); the
        // same rule as gemm_planes_kernel (mdbn_planes.hip, PL_STAGE_ACC), so that its 32x32x16 form stays bit-identical.
        constexpr bool STAGE_ACC = X6_STAGE_ACC && LA == LAY_K && LB == LAY_MN;
#define X6_CHAIN(T, FA, FB, a, b)                                                                 \
    do {                                                                                          \
        if constexpr (AP == 3) {                    /* smallest products first */                 \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2][a], FB[0][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[1][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[0][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);          \
        } else {        /* A = a1 exactly: a1 b3 + a1 b2 + a1 b1 is the full product */           \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);          \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);          \
        }                                                                                         \
    } while (0)
#define X6_MMA(FA, FB)                                                                            \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                 \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                           \
            if constexpr (STAGE_ACC) {                                                            \
                f32x16 t_;                                                                        \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) t_[e] = 0.f;                       \
                X6_CHAIN(t_, FA, FB, a, b);                                                       \
                acc[a][b] += t_;                                                                  \
                __builtin_amdgcn_sched_barrier(0);  /* one step-local accumulator alive at a time (hipcc otherwise hoists all four chains and spills) */ \
            } else {                                                                              \
                X6_CHAIN(acc[a][b], FA, FB, a, b);                                                \
            }                                                                                     \
        }
        __syncthreads();                             // slice 0 staged
        X6_FRAGS(f0a, f0b, lds, 0);
        for (int it = 0; it < nt; ++it) {
            const unsigned char* base = lds + (it & 1) * X6_BUF;
            const unsigned char* next = lds + ((it + 1) & 1) * X6_BUF;
            STAMP(0);
            X6_FRAGS(f1a, f1b, base, 1);
            X6_MMA(f0a, f0b);
            STAMP(1);
            // Pin the F1 reads BEFORE the barrier: hipcc may sink LDS loads whose only use sits in a later basic
            // block past s_barrier (an IntrNoMem intrinsic; the workgroup fences around it lower to waits only).
            // That is what the "guarded reload" build did -- the F1 reads of slice `it` ended up after the barrier
            // that lets the producers refill its buffer (root cause).  An empty asm that consumes the
            // registers costs nothing and makes the position of the loads a data dependence.
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (pl < AP) asm volatile("" :: "v"(f1a[pl][a]));
                    asm volatile("" :: "v"(f1b[pl][a]));
                }
            __syncthreads();                         // (waits for F1: every read of this slice's buffer is done)
            STAMP(2);
            X6_FRAGS(f0a, f0b, next, 0);
            X6_MMA(f1a, f1b);
            STAMP(3);
        }
#undef X6_FRAGS
#undef X6_MMA
#undef X6_CHAIN
        // accumulator (32x32): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        if constexpr (FUSED != 0) {
            constexpr int LDT = BN + 8;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        smem[(wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * q) * LDT + wn + 32 * b + i] = acc[a][b][e];
        } else {
            float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = n0 + wn + 32 * b + i;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * q;
                        if (!RAGGED || interior || (row < g.M && col < g.Nst)) C[(int64_t)row * g.ldc + col] = acc[a][b][e];
                    }
                }
            return;
        }
    }
    if constexpr (FUSED != 0) {
        __syncthreads();
        if constexpr (FUSED == 1) fused_tile_epilogue<BM, BN, 64 * (4 + 2 * PW)>(g.epi, smem, m0, n0);   // activation + sampling
        else fused_update_epilogue<BM, BN, 64 * (4 + 2 * PW)>(g.upd, smem, m0, n0);                       // parameter update
    }
}

template <int LA, int LB, int FUSED, int AP, bool RAGGED, int PW>
static hipError_t launch_bf16x6_p(const GemmArgs& g, hipStream_t s)
{
    constexpr int park = (128 * (128 + 8) + 8) * (int)sizeof(float);
    constexpr int lds_bytes = 2 * X6_BUF > park ? 2 * X6_BUF : park;
    static bool attr_set = false;
    auto kern = gemm_bf16x6_kernel<LA, LB, FUSED, AP, RAGGED, PW>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.splitk), dim3(64 * (4 + 2 * PW)), lds_bytes, s, g);
    return hipGetLastError();
}

template <int LA, int LB, int FUSED, int AP, bool RAGGED>
static hipError_t launch_bf16x6_r(const GemmArgs& g, hipStream_t s)
{
    // the step-local accumulators of the propup layout (X6_STAGE_ACC) need 187 VGPRs with six products: that does not fit the
    // 168-register budget of a 12-wave block, so those instantiations run with two producer waves per operand
    if constexpr (X6_STAGE_ACC && LA == LAY_K && LB == LAY_MN && AP == 3) return launch_bf16x6_p<LA, LB, FUSED, AP, RAGGED, 2>(g, s);
    else
    return g.x6_pw == 4 ? launch_bf16x6_p<LA, LB, FUSED, AP, RAGGED, 4>(g, s) : launch_bf16x6_p<LA, LB, FUSED, AP, RAGGED, 2>(g, s);
}

template <int LA, int LB, int FUSED, int AP = 3>
static hipError_t launch_bf16x6_t(const GemmArgs& g, hipStream_t s)
{
    const bool whole = g.M % 128 == 0 && g.N % 128 == 0 && g.K % X6_KB == 0 && (g.splitk == 1 || g.kchunk * g.splitk == g.K);
    return whole ? launch_bf16x6_r<LA, LB, FUSED, AP, false>(g, s) : launch_bf16x6_r<LA, LB, FUSED, AP, true>(g, s);
}

hipError_t launch_gemm_bf16x6(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (g.kchunk % X6_KB || g.tiles_m != (g.M + 127) / 128 || g.tiles_n != (g.N + 127) / 128 || (g.lda & 3) ||
        (g.ldb & 3) || (g.fused && g.splitk != 1))
        return hipErrorInvalidValue;
    if (la == LAY_MN && lb == LAY_MN && g.fused == 2) return launch_bf16x6_t<LAY_MN, LAY_MN, 2>(g, s);
    if (la == LAY_MN && lb == LAY_MN && g.fused == 0) return launch_bf16x6_t<LAY_MN, LAY_MN, 0>(g, s);
    if (la == LAY_K && lb == LAY_K && g.fused == 1 && g.x6 == 2) return launch_bf16x6_t<LAY_K, LAY_K, 1, 1>(g, s);
    if (la == LAY_K && lb == LAY_MN && g.fused == 1 && g.x6 == 2) return launch_bf16x6_t<LAY_K, LAY_MN, 1, 1>(g, s);
    if (la == LAY_K && lb == LAY_K && g.fused == 1) return launch_bf16x6_t<LAY_K, LAY_K, 1>(g, s);
    if (la == LAY_K && lb == LAY_MN && g.fused == 1) return launch_bf16x6_t<LAY_K, LAY_MN, 1>(g, s);
    if (la == LAY_K && lb == LAY_K && g.fused == 0 && g.x6 == 2) return launch_bf16x6_t<LAY_K, LAY_K, 0, 1>(g, s);
    if (la == LAY_K && lb == LAY_MN && g.fused == 0 && g.x6 == 2) return launch_bf16x6_t<LAY_K, LAY_MN, 0, 1>(g, s);
    if (la == LAY_K && lb == LAY_K && g.fused == 0) return launch_bf16x6_t<LAY_K, LAY_K, 0>(g, s);
    if (la == LAY_K && lb == LAY_MN && g.fused == 0) return launch_bf16x6_t<LAY_K, LAY_MN, 0>(g, s);
    return hipErrorInvalidValue;
}

hipError_t launch_skinny_gemm(int la, int lb, const GemmArgs& g, hipStream_t s);

hipError_t launch_stream_gemm(int la, int lb, const GemmArgs& g, hipStream_t s);     // mdbn_stream.hip

hipError_t launch_gemm(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (g.skinny && g.x6) return launch_stream_gemm(la, lb, g, s);
    if (g.x6) return launch_gemm_bf16x6(la, lb, g, s);
    if (g.skinny) return launch_skinny_gemm(la, lb, g, s);
    if (g.fused) {      // forward passes fuse the activation, the statistics GEMM the update (separate
                        // instantiations: the unfused kernels keep their register allocation)
        if (g.splitk != 1) return hipErrorInvalidValue;
        if (g.fused == 1 && la == LAY_K && lb == LAY_MN) return launch_gemm_l<LAY_K, LAY_MN, 1>(g, s);
        if (g.fused == 1 && la == LAY_K && lb == LAY_K) return launch_gemm_l<LAY_K, LAY_K, 1>(g, s);
        if (g.fused == 2 && la == LAY_MN && lb == LAY_MN) return launch_gemm_l<LAY_MN, LAY_MN, 2>(g, s);
        return hipErrorInvalidValue;
    }
    if (la == LAY_K && lb == LAY_MN) return launch_gemm_l<LAY_K, LAY_MN, 0>(g, s);
    if (la == LAY_K && lb == LAY_K) return launch_gemm_l<LAY_K, LAY_K, 0>(g, s);
    if (la == LAY_MN && lb == LAY_MN) return launch_gemm_l<LAY_MN, LAY_MN, 0>(g, s);
    return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------------
// Register-streaming ("skinny") GEMM: no LDS staging, operands go from global memory / L2
// straight into the MFMA operand registers.  Two regimes use it:
//  * minibatches of <= 64 rows (the reference trains with batch_size 20, dbn.py / MDBN.py): at
//    M <= 32 a weight is used by exactly ONE MFMA, so staging W through LDS buys nothing and a
//    128-row tile wastes >= 75% of the matrix pipe on zero rows (W then streams at 1/4 of the HBM
//    rate).  MFMA issue at M <= 32 equals ~8.6 TB/s of W: the pass is HBM / latency bound, as a
//    batched GEMV should be.
//  * small layers at any batch size (operands L2-resident, too few 128x128 tiles to fill the chip
//    without split-K): one launch with a fused epilogue replaces GEMM + slabs + epilogue kernel.
// A block owns a (32*MI)-row x 32-column output tile and a K range; its 8 waves take interleaved
// K-octets, keep one batch of 4 octets of loads in flight ahead of the MFMAs (two register sets),
// and reduce their partial accumulators through LDS in wave order (deterministic).  Epilogues on
// the reduced tile: none (split-K slab / plain C), FUSED 1 = bias + activation + sampling
// (act_quad), FUSED 2 = the parameter update (statistics GEMM, update_rule4).
//   LAY_K  operand X[rows][ld], K contiguous : lane (i, h) <- X[r0 + i][k8 + 4h .. +3]  (one float4)
//   LAY_MN operand X[K][ld], rows contiguous : lane (i, h) <- X[k8 + 4h + t][r0 + i], t < 4
// MFMA t of an octet multiplies the k-pair {k8 + t, k8 + 4 + t}.  Rows >= M / columns >= N read a
// clamped (valid) address: they only reach accumulator rows / columns that are never stored.  The
// K tail is zero-filled in both operands.
// ----------------------------------------------------------------------------------
constexpr int SKINNY_U = 4;

template <int MI>
struct SkinnyRegs {
    float4 a[SKINNY_U][MI];
    float4 b[SKINNY_U];
};

template <int LAY>
__device__ __forceinline__ float4 skinny_load_operand(const float* p, int64_t ld, int k, int kend, bool full)
{
    float4 r;
    if (full) {
        if (LAY == LAY_K) {
            r = *reinterpret_cast<const float4*>(p + k);
        } else {
            const float* q = p + (int64_t)k * ld;
            r.x = q[0]; r.y = q[ld]; r.z = q[2 * ld]; r.w = q[3 * ld];
        }
    } else {        // K tail: component guards, zero fill
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = k + t < kend;
            if (LAY == LAY_K) v[t] = ok ? p[k + t] : 0.f;
            else v[t] = ok ? p[(int64_t)(k + t) * ld] : 0.f;
        }
        r = make_float4(v[0], v[1], v[2], v[3]);
    }
    return r;
}

template <int LA, int LB, int MI, int FUSED>
__global__ __launch_bounds__(64 * SKINNY_WAVES) void skinny_gemm_kernel(GemmArgs g)
{
    MDBN_GEMM_ARGS_EARLY(g);
    constexpr int NW = SKINNY_WAVES, U = SKINNY_U, LDT = SKINNY_LDT, BM = 32 * MI;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [NW][BM][LDT] (+ 8)
    // block -> (K range, row tile, 32-column strip), strips fastest
    const int per_split = g.tiles_n * g.tiles_m;
    const int ks = blockIdx.x / per_split, rem = blockIdx.x - ks * per_split;
    const int tm = rem / g.tiles_n, st = rem - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = st * 32;
    const int kbeg = ks * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int noct = (kend - kbeg + 7) >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;

    const float* aptr[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        const int am = min(m0 + i + 32 * a, g.M - 1);
        aptr[a] = LA == LAY_K ? g.A + (int64_t)am * g.lda : g.A + am;
    }
    const int bn = min(n0 + i, g.N - 1);
    const float* bptr = LB == LAY_K ? g.B + (int64_t)bn * g.ldb : g.B + bn;

    f32x16 acc[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

    if constexpr (LA == LAY_MN && LB == LAY_MN) {
        // statistics GEMM: the bias statistics / cost total / bias update units (finalize_unit: the same functions and order
        // as finalize_stats_kernel) spread over the waves of all blocks, ahead of the operand stream -- one launch fewer
        if (g.fin_enabled) {
            const int nu = fin_units(g.fin);
            for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += NW * (int)gridDim.x)
                finalize_unit(g.fin, unit, lane);
        }
    }

    // batch = U octets of this wave (octets wave, wave + NW, ...); two register sets
    SkinnyRegs<MI> r0, r1;
#define SKINNY_LOAD(R, OB)                                                                   \
    do {                                                                                     \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
            const int o = (OB) + u * NW;                                                     \
            if (o < noct) {                                                                  \
                const int k8 = kbeg + 8 * o;                                                 \
                const bool full = k8 + 8 <= kend;                                            \
                _Pragma("unroll") for (int a = 0; a < MI; ++a)                               \
                    R.a[u][a] = skinny_load_operand<LA>(aptr[a], g.lda, k8 + 4 * h, kend, full); \
                R.b[u] = skinny_load_operand<LB>(bptr, g.ldb, k8 + 4 * h, kend, full);       \
            }                                                                                \
        }                                                                                    \
    } while (0)
#define SKINNY_MMA(R, OB)                                                                    \
    do {                                                                                     \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                      \
            if ((OB) + u * NW < noct) {                                                      \
                _Pragma("unroll") for (int a = 0; a < MI; ++a) {                             \
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[u][a].x, R.b[u].x, acc[a], 0, 0, 0); \
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[u][a].y, R.b[u].y, acc[a], 0, 0, 0); \
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[u][a].z, R.b[u].z, acc[a], 0, 0, 0); \
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.a[u][a].w, R.b[u].w, acc[a], 0, 0, 0); \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
    } while (0)

    constexpr int STEP = NW * U;
    SKINNY_LOAD(r0, wave);
    for (int ob = wave; ob < noct; ob += 2 * STEP) {
        SKINNY_LOAD(r1, ob + STEP);
        SKINNY_MMA(r0, ob);
        SKINNY_LOAD(r0, ob + 2 * STEP);
        SKINNY_MMA(r1, ob + STEP);
    }
#undef SKINNY_LOAD
#undef SKINNY_MMA

    skinny_tile_epilogue<MI, FUSED, false>(g, acc, smem, ks, m0, n0, (int)blockIdx.x);
}

template <int LA, int LB, int MI, int FUSED>
static hipError_t launch_skinny_t(const GemmArgs& g, hipStream_t s)
{
    constexpr int lds_bytes = (SKINNY_WAVES * 32 * MI * SKINNY_LDT + 8) * (int)sizeof(float);
    static bool attr_set = false;
    auto kern = skinny_gemm_kernel<LA, LB, MI, FUSED>;
    if (!attr_set && lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_n * g.tiles_m * g.splitk), dim3(64 * SKINNY_WAVES), lds_bytes, s, g);
    return hipGetLastError();
}

hipError_t launch_skinny_gemm(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (g.M < 1 || (g.mi != 1 && g.mi != 2) || (int64_t)g.tiles_m * 32 * g.mi < g.M || g.kchunk % 8 != 0 ||
        (g.fused && g.splitk != 1))
        return hipErrorInvalidValue;
#define SKINNY_CASE(LAV, LBV, MIV, FV) \
    if (la == LAV && lb == LBV && g.mi == MIV && g.fused == FV) return launch_skinny_t<LAV, LBV, MIV, FV>(g, s)
    // forward passes: plain / activation epilogue
    SKINNY_CASE(LAY_K, LAY_K, 1, 0);  SKINNY_CASE(LAY_K, LAY_K, 1, 1);
    SKINNY_CASE(LAY_K, LAY_K, 2, 0);  SKINNY_CASE(LAY_K, LAY_K, 2, 1);
    SKINNY_CASE(LAY_K, LAY_MN, 1, 0); SKINNY_CASE(LAY_K, LAY_MN, 1, 1);
    SKINNY_CASE(LAY_K, LAY_MN, 2, 0); SKINNY_CASE(LAY_K, LAY_MN, 2, 1);
    // statistics GEMM of small layers: plain / parameter update
    SKINNY_CASE(LAY_MN, LAY_MN, 2, 0); SKINNY_CASE(LAY_MN, LAY_MN, 2, 2);
#undef SKINNY_CASE
    return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------------
// split-K reduce + bias + activation + sampling epilogue.
//   gauss = 0: pre = sum + bias ; mean = sigmoid(pre) ; sample = (u < mean)
//              (propup/sample_h_given_v rbm.py:198-213, propdown/sample_v_given_h :226-240)
//   gauss = 1: mean = pre = sum + bias ; sample = mean + N(0,1)      (GRBM rbm.py:650-658)
//   cost (target != NULL): sum of BCE(sigmoid(pre), target) (rbm.py:479-480) or of
//              (sigmoid(mean) - target)^2 (rbm.py:697), one partial per block.
//   colsum != NULL: per-thread partial of the bias statistics (rbm.py:416-417) over the
//              thread's 4 rows: colsum_kind 0 = sum of the stored (scaled) mean,
//              1 = sum of (target - mean), 2 = sum of (target - sample); written to
//              colsum[row_group][col].
// One thread = 4 rows x CW columns (CW = 4, 2 or 1: float4 / float2 / float traffic), one
// Philox block per column.  The kernel is latency-bound (one wave of blocks, slabs resident in
// L2 / Infinity Cache), so narrower threads = more of them = shorter serial chain per thread.
// NS > 0: compile-time split count (all slab loads of a row in flight together).
// ----------------------------------------------------------------------------------
template <int CW> struct VecIO;
template <> struct VecIO<4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct VecIO<2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2]) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct VecIO<1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};

// MODE >= 0: e.gauss (bit 1) and "a sample is wanted" (bit 0) as compile-time facts (the launcher checks them): the
// executed path is then short straight-line code.  With every case compiled into one body the activation arithmetic of a
// tile took three times as long (measured on the fused form of this epilogue, mdbn_device.h fused_tile_epilogue_4x4).
template <int NS, int CW, int MODE = -1>
__global__ __launch_bounds__(256) void act_epilogue_kernel(EpiArgs e)
{
    __shared__ float red[4];
    // the arguments needed before the slab loads, requested in ONE batch (hipcc loads kernarg fields where they are first used:
    // three scalar-load round trips in a row at the head of a 6-us launch)
    asm volatile("" :: "s"(e.slabs), "s"(e.slab_stride), "s"(e.nsplit), "s"(e.rows), "s"(e.cols), "s"(e.ld), "s"(e.bias),
                 "s"(e.target), "s"(e.ld_target), "s"(e.target_idx), "s"(e.mean_planes), "s"(e.plane_stride), "s"(e.sample_plane),
                 "s"(e.colsum), "s"(e.target_idx64), "s"((int)blockDim.x));
    const bool is_gauss = MODE < 0 ? e.gauss != 0 : (MODE & 2) != 0;
    const int ldc = (int)(e.ld / CW);                    // column groups per row
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int rg = (int)(idx / ldc), cq = (int)(idx - (int64_t)rg * ldc);
    const int r0 = rg * 4, c0 = cq * CW;
    float cost = 0.f;
    if (r0 < e.rows) {
        float bias[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) bias[j] = c0 + j < e.cols ? e.bias[c0 + j] : 0.f;
        float pre[4][CW];
        const float* base = e.slabs + (int64_t)r0 * e.ld + c0;
        // the split-K partials are summed in float64 and rounded to float32 ONCE (a single slab passes through unchanged,
        // so the one-slab form stays bitwise the fused epilogues); the bias is then added in float32 as everywhere else
        double acc[4][CW];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CW; ++j) acc[r][j] = 0.0;
        // Every load of a batch is issued before the first addition: the addresses are clamped into the slabs (a dead row
        // re-reads the last live one, a dead slab the last slab) and the dead values become an exact 0.0 by a select
        // afterwards.  Predicated loads -- `if (row < rows) load` -- put each load into its own basic block, and with the
        // float64 accumulation hipcc then converted and added right behind every single load: one memory round trip per
        // slab and row (the generic body: 49 us for the 37 slabs of 19 937 -> 400 at batch 20, 12 us before the float64
        // sums; 17 us per launch at 2048 -> 400), and four round trips instead of one in the unrolled bodies.
        const int rlast = e.rows - 1 - r0;                   // last live row of this thread's four (>= 0)
        // the cost targets of the four rows (index -> row: two dependent loads each) requested up front, not row by row
        // inside the activation loop (eight round trips in a row: most of this launch's time on the last visible pass)
        float tgt4[4][CW];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CW; ++j) tgt4[r][j] = 0.f;
        if (e.target) {
            int64_t srow[4];
            epi_target_rows4(e, r0, srow);
#pragma unroll
            for (int r = 0; r < 4; ++r) VecIO<CW>::load(e.target + srow[r] * e.ld_target + c0, tgt4[r]);
        }
        if (NS > 0) {
            constexpr int RB = NS * CW >= 64 ? 1 : (NS * CW >= 32 ? 2 : 4);      // rows per batch: <= 64 floats in flight
#pragma unroll
            for (int rb = 0; rb < 4; rb += RB) {
                float v[RB][NS > 0 ? NS : 1][CW];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx)
                        VecIO<CW>::load(base + (int64_t)min(rb + r, rlast) * e.ld + (int64_t)sidx * e.slab_stride, v[r][sidx]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
                        for (int j = 0; j < CW; ++j) acc[rb + r][j] += rb + r <= rlast ? (double)v[r][sidx][j] : 0.0;
            }
        } else {
            // any split count: 16 / CW slabs x 4 rows of loads in flight per round, summed in slab order
            constexpr int SB = 16 / CW;               // slabs per round: 64 loaded floats per thread in flight
            int nsplit = e.nsplit;
            if (e.bal_P) {      // slabs of a balanced GEMM: this 128x128 tile has one per workgroup that shared its stages
                const int tm = r0 >> 7, tn = c0 >> 7;
                const int t = e.bal_tiles_m <= e.bal_tiles_n ? tn * e.bal_tiles_m + tm : tm * e.bal_tiles_n + tn;
                nsplit = bal_tile_slabs(t, e.bal_tiles_m * e.bal_tiles_n, e.bal_S, e.bal_P);
            }
            for (int s0 = 0; s0 < nsplit; s0 += SB) {
                float v[4][SB][CW];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < SB; ++u)
                        VecIO<CW>::load(base + (int64_t)min(r, rlast) * e.ld + (int64_t)min(s0 + u, nsplit - 1) * e.slab_stride, v[r][u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int u = 0; u < SB; ++u)
#pragma unroll
                        for (int j = 0; j < CW; ++j) acc[r][j] += (r <= rlast && s0 + u < nsplit) ? (double)v[r][u][j] : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < CW; ++j) pre[r][j] = (float)acc[r][j] + bias[j];
        uint32_t wa[CW][4], wb[CW][4];       // [col][row]
        const bool need_u = MODE < 0 ? (e.sample != nullptr || e.sample_plane != nullptr) : (MODE & 1) != 0;
        const bool need_z = need_u && is_gauss;
        if (need_u) {
            const uint64_t g0 = e.rng.row_offset + (uint64_t)r0;
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                philox_rows4(e.rng, e.rng.draw, g0, (uint32_t)(c0 + j), wa[j]);
                if (need_z) philox_rows4(e.rng, e.rng.draw | MDBN_NORMAL_BIT, g0, (uint32_t)(c0 + j), wb[j]);
            }
        }
        float csum[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) csum[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r0 + r >= e.rows) continue;
            const int64_t off = (int64_t)(r0 + r) * e.ld + c0;
            float mean[CW], samp[CW], tgt[CW];
#pragma unroll
            for (int j = 0; j < CW; ++j) tgt[j] = tgt4[r][j];
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const bool live = c0 + j < e.cols;
                const float x = pre[r][j];
                float m, sv = 0.f;
                if (is_gauss) {
                    m = x;
                    if (need_u) {
                        const float u1 = philox_u01(wa[j][r]), u2 = philox_u01(wb[j][r]);
                        sv = m + sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    }
                } else {
                    m = sigmoidf_(x);
                    if (need_u) sv = philox_u01(wa[j][r]) < m ? 1.0f : 0.0f;
                }
                if (e.target && live) {
                    if (is_gauss) { const float d = sigmoidf_(x) - tgt[j]; cost += d * d; }
                    else cost += tgt[j] * softplusf_(-x) + (1.0f - tgt[j]) * softplusf_(x);
                }
                if (!live) { m = 0.f; sv = 0.f; pre[r][j] = 0.f; }   // keep pad columns zero
                const float ms = m * e.mean_scale;
                mean[j] = ms;
                samp[j] = sv;
                if (live) csum[j] += e.colsum_kind == 0 ? ms : (e.colsum_kind == 1 ? tgt[j] - m : tgt[j] - sv);
            }
            if (e.pre) VecIO<CW>::store(e.pre + off, pre[r]);
            if (e.mean) VecIO<CW>::store(e.mean + off, mean);
            if (e.sample) VecIO<CW>::store(e.sample + off, samp);
            if (e.mean_planes || e.sample_plane) {      // CW bf16 values per plane and row: one packed store each
                unsigned short q[4][CW];                 // [p1, p2, p3, sample][column]
#pragma unroll
                for (int j = 0; j < CW; ++j) {
                    split3(mean[j], q[0][j], q[1][j], q[2][j]);
                    q[3][j] = (unsigned short)(__builtin_bit_cast(unsigned, samp[j]) >> 16);
                }
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) {
                    unsigned short* dst = pl < 3 ? (e.mean_planes ? e.mean_planes + pl * e.plane_stride + off : nullptr)
                                                 : (e.sample_plane ? e.sample_plane + off : nullptr);
                    if (!dst) continue;
                    if constexpr (CW == 4) {
                        uint2 w; w.x = q[pl][0] | ((unsigned)q[pl][1] << 16); w.y = q[pl][2] | ((unsigned)q[pl][3] << 16);
                        *reinterpret_cast<uint2*>(dst) = w;
                    } else if constexpr (CW == 2) {
                        *reinterpret_cast<unsigned*>(dst) = q[pl][0] | ((unsigned)q[pl][1] << 16);
                    } else {
                        dst[0] = q[pl][0];
                    }
                }
            }
        }
        if (e.colsum) VecIO<CW>::store(e.colsum + (int64_t)rg * e.ld + c0, csum);
    }
    if (e.cost_partials) {
        const float tot = block_sum(cost, red);
        if (threadIdx.x == 0) e.cost_partials[blockIdx.x] = tot;
    }
}

static int g_epilogue_cw = 0;     // 0 = auto; 4, 2 or 1 (mdbn_set_option "epilogue_cw")
void set_epilogue_cw(int cw) { g_epilogue_cw = cw; }

int epilogue_cw(int64_t rows, int64_t ld)
{
    if (g_epilogue_cw) return g_epilogue_cw;
    // enough 4-wide threads to fill the chip several times over? then keep float4 traffic;
    // small outputs take narrower threads so that more CUs share the slab reads
    const int64_t quads = (rows + 3) / 4;
    if (quads * (ld / 4) >= 8 * 256 * 256) return 4;
    return quads * (ld / 2) >= 64 * 256 ? 2 : 1;
}

// threads per block: a small output (a 20-row minibatch) is spread over 64-thread blocks -- with
// 256-thread blocks 4 CUs would read all the split-K slabs (12 us for 37 slabs of 20 x 400)
static int g_epilogue_threads = 0;     // 0 = auto; 64 | 128 | 256 (mdbn_set_option "epilogue_threads")
void set_epilogue_threads(int t) { g_epilogue_threads = t; }
static int epilogue_threads(int64_t rows, int64_t ld)
{
    if (g_epilogue_threads) return g_epilogue_threads;
    const int64_t n = ((rows + 3) / 4) * (ld / epilogue_cw(rows, ld));
    return n <= 64 * 256 ? 64 : 256;
}

int epilogue_blocks(int64_t rows, int64_t ld)
{
    const int cw = epilogue_cw(rows, ld), t = epilogue_threads(rows, ld);
    const int64_t n = ((rows + 3) / 4) * (ld / cw);
    return (int)((n + t - 1) / t);
}

template <int CW>
static void launch_act_epilogue_cw(const EpiArgs& e, hipStream_t s)
{
    const int64_t n = ((int64_t)(e.rows + 3) / 4) * (e.ld / CW);
    const int t = epilogue_threads(e.rows, e.ld);
    const dim3 grid((unsigned)((n + t - 1) / t)), block(t);
    // MODE-specialised bodies (the activation arithmetic as straight-line code: a third of the generic body's time) for the
    // split counts the plans produce -- 8 / 2 (headline), 16 / 3 / 4 (ragged and mid-size layers: 2048 -> 400 splits propup 16
    // ways and propdown 3; its eleven epilogue launches per CD-5 step ran the generic body at 17 us each, 48 % of the step's
    // GPU time, profiles/r04zk_ge_2048_400_cd5_kernel_stats.csv)
    const int mode = (e.gauss ? 2 : 0) | ((e.sample != nullptr || e.sample_plane != nullptr) ? 1 : 0);
    if (mode != 3 && (e.bal_P || (e.nsplit != 6 && e.nsplit != 7))) {     // (6 / 7: the unrolled bodies of the balanced launches below)
        const int ns = e.bal_P ? 0 : (e.nsplit == 8 || e.nsplit == 2 || e.nsplit == 16 || e.nsplit == 3 || e.nsplit == 4) ? e.nsplit : 0;
#define EPI_MODE_CASE(NSV, MV) \
    if (ns == NSV && mode == MV) { hipLaunchKernelGGL((act_epilogue_kernel<NSV, CW, MV>), grid, block, 0, s, e); return; }
#define EPI_MODE_CASES(NSV) EPI_MODE_CASE(NSV, 0) EPI_MODE_CASE(NSV, 1) EPI_MODE_CASE(NSV, 2)
        EPI_MODE_CASES(8) EPI_MODE_CASES(2) EPI_MODE_CASES(16) EPI_MODE_CASES(3) EPI_MODE_CASES(4)
        // (NOT the rolled loop of any other count, NS = 0: its specialised bodies came out FOUR times slower than the generic
        //  one -- 49 instead of 12 us for the 37 slabs of 19 937 -> 400 at batch 20, profiles/r04zo_ge_19937_400_b20_kernel_stats.csv)
#undef EPI_MODE_CASES
#undef EPI_MODE_CASE
    }
    switch (e.bal_P ? 0 : e.nsplit) {       // (a Gaussian unit WITH a sample: the generic body)
        case 1: hipLaunchKernelGGL((act_epilogue_kernel<1, CW>), grid, block, 0, s, e); break;
        case 2: hipLaunchKernelGGL((act_epilogue_kernel<2, CW>), grid, block, 0, s, e); break;
        case 4: hipLaunchKernelGGL((act_epilogue_kernel<4, CW>), grid, block, 0, s, e); break;
        case 8: hipLaunchKernelGGL((act_epilogue_kernel<8, CW>), grid, block, 0, s, e); break;
        case 7: hipLaunchKernelGGL((act_epilogue_kernel<7, CW>), grid, block, 0, s, e); break;      // balanced propup at P = 224
        case 6: hipLaunchKernelGGL((act_epilogue_kernel<6, CW>), grid, block, 0, s, e); break;      // ... at P = 192
        default: hipLaunchKernelGGL((act_epilogue_kernel<0, CW>), grid, block, 0, s, e); break;
    }
}

hipError_t launch_act_epilogue(const EpiArgs& e, hipStream_t s)
{
    switch (epilogue_cw(e.rows, e.ld)) {
        case 4: launch_act_epilogue_cw<4>(e, s); break;
        case 2: launch_act_epilogue_cw<2>(e, s); break;
        default: launch_act_epilogue_cw<1>(e, s); break;
    }
    return hipGetLastError();
}

// out = sum over slabs (plain split-K combine; used when the statistic GEMM is split)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, int nsplit,
                                                        int64_t slab_stride, int64_t n4,
                                                        float* __restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nsplit; s0 += 8) {      // 8 loads in flight, summed in slab order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = s0 + u < nsplit ? reinterpret_cast<const float4*>(slabs + (int64_t)(s0 + u) * slab_stride)[i]
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

hipError_t launch_sum_slabs(const float* slabs, int nsplit, int64_t slab_stride, int64_t n,
                            float* out, hipStream_t s)
{
    const int64_t n4 = n >> 2;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid), dim3(256), 0, s, slabs, nsplit, slab_stride, n4, out);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// minibatch gather  train_set_x[indexes]  (dbn.py:307, rbm.py:538)
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t n_rows,
                                                          int64_t ld_src, const void* __restrict__ idx,
                                                          int idx64, int64_t ld4, float* __restrict__ dst,
                                                          int64_t ld_dst)
{
    // one float4 per thread: grid = (column chunks of 256 float4, rows)
    const int64_t r = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld4) return;
    int64_t s = r;
    if (idx) s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
    if (s < 0) s += n_rows;                       // numpy-style negative index
    s = s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);   // never fault on a bad index
    reinterpret_cast<float4*>(dst + r * ld_dst)[c] = reinterpret_cast<const float4*>(src + s * ld_src)[c];
}

hipError_t launch_gather(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src,
                         const void* idx, int idx64, int64_t n_idx, float* dst, int64_t ld_dst,
                         hipStream_t s)
{
    if (n_idx <= 0) return hipSuccess;
    const int64_t ld4 = cols_ld >> 2;
    const int64_t rows_per_launch = 65535;        // gridDim.y limit
    for (int64_t r0 = 0; r0 < n_idx; r0 += rows_per_launch) {
        const int64_t nr = std::min(rows_per_launch, n_idx - r0);
        const void* ip = idx ? (idx64 ? (const void*)(reinterpret_cast<const int64_t*>(idx) + r0)
                                      : (const void*)(reinterpret_cast<const int32_t*>(idx) + r0))
                             : nullptr;
        // identity gather of a later chunk: offset the source instead of the (absent) index list
        const float* sp = idx ? src : src + r0 * ld_src;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ld4 + 255) / 256), (unsigned)nr), dim3(256), 0, s,
                           sp, idx ? n_rows : n_rows - r0, ld_src, ip, idx64, ld4, dst + r0 * ld_dst, ld_dst);
    }
    return hipGetLastError();
}

// The same gather with a SMALL footprint, for sources in pinned host memory read over PCIe beside a running CD step
// (shared.HostTable, StepFunction.prefetch): `blocks` workgroups (<= 64) of 256 threads walk the rows, four 16-byte loads
// in flight per thread (64 KB per row pass of a block: more than enough to cover the ~2 us PCIe round trip at 55 GB/s),
// at most 40 VGPRs -- so its waves fit beside a resident 144-KB GEMM workgroup (2 x 232 VGPRs per SIMD) instead of
// filling every wave slot of the chip for the 150 us the transfer takes (the wide kernel above serialised with the step:
// 321 us per step against 146 resident, profiles/r03g_bench.json).
__global__ __launch_bounds__(256) void gather_rows_slim_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src,
                                                               const void* __restrict__ idx, int idx64, int64_t ld4,
                                                               int64_t n_idx, float* __restrict__ dst, int64_t ld_dst)
{
    const int nt = blockDim.x;
    for (int64_t r = blockIdx.x; r < n_idx; r += gridDim.x) {
        int64_t s = r;
        if (idx) s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
        if (s < 0) s += n_rows;
        s = s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);
        const float4* in = reinterpret_cast<const float4*>(src + s * ld_src);
        float4* out = reinterpret_cast<float4*>(dst + r * ld_dst);
        for (int64_t c0 = threadIdx.x; c0 < ld4; c0 += 4 * nt) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u * nt < ld4) v[u] = in[c0 + u * nt];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u * nt < ld4) out[c0 + u * nt] = v[u];
        }
    }
}

hipError_t launch_gather_slim(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src, const void* idx, int idx64,
                              int64_t n_idx, float* dst, int64_t ld_dst, int blocks, int threads, hipStream_t s)
{
    if (n_idx <= 0) return hipSuccess;
    blocks = (int)std::min<int64_t>(std::max(1, std::min(blocks, 1024)), n_idx);
    hipLaunchKernelGGL(gather_rows_slim_kernel, dim3(blocks), dim3(threads), 0, s, src, n_rows, ld_src, idx, idx64, cols_ld >> 2,
                       n_idx, dst, ld_dst);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// bias statistics / cost total / bias update as a kernel of its own (finalize_unit above): one unit
// per wave.  The single-device fused step runs the same units inside the statistics GEMM instead.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_stats_kernel(FinArgs f)
{
    finalize_unit(f, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), threadIdx.x & 63);
}

FinArgs make_fin_args(const float* posP, const float* negP, const float* partV, int ngroups, int64_t ldh, int64_t ldv,
                      const float* cost_partials, int n_cost, float* s_h, float* s_v, float* cost,
                      const BiasUpd* bias_update)
{
    FinArgs f;
    memset(&f, 0, sizeof f);
    f.posP = posP; f.negP = negP; f.partV = partV; f.ngroups = ngroups; f.ldh = ldh; f.ldv = ldv;
    f.cost_partials = cost_partials; f.n_cost = n_cost; f.s_h = s_h; f.s_v = s_v; f.cost = cost;
    if (bias_update) { f.bu = *bias_update; f.do_bias = 1; }
    return f;
}

hipError_t launch_finalize_stats(const float* posP, const float* negP, const float* partV, int ngroups,
                                 int64_t ldh, int64_t ldv, const float* cost_partials, int n_cost,
                                 float* s_h, float* s_v, float* cost, const BiasUpd* bias_update, hipStream_t s)
{
    const FinArgs f = make_fin_args(posP, negP, partV, ngroups, ldh, ldv, cost_partials, n_cost, s_h, s_v, cost,
                                    bias_update);
    const int units = (int)((ldh + ldv + 15) / 16) + 1;         // + the cost unit
    hipLaunchKernelGGL(finalize_stats_kernel, dim3((units + 3) / 4), dim3(256), 0, s, f);
    return hipGetLastError();
}

// column sums over 4-row groups of caller-supplied buffers (mdbn_cd_stats):
// out[g][col] = sum_{r in group g} (X[r][col] - (Y ? Y[r][col] : 0))
__global__ __launch_bounds__(256) void colsum_groups_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                            int rows, int64_t ld, float* __restrict__ out)
{
    const int ld4 = (int)(ld >> 2);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int rg = (int)(idx / ld4), cq = (int)(idx - (int64_t)rg * ld4);
    if (rg * 4 >= rows) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = rg * 4; r < min(rows, rg * 4 + 4); ++r) {
        const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)r * ld + cq * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        if (Y) {
            const float4 y = *reinterpret_cast<const float4*>(Y + (int64_t)r * ld + cq * 4);
            a.x -= y.x; a.y -= y.y; a.z -= y.z; a.w -= y.w;
        }
    }
    *reinterpret_cast<float4*>(out + (int64_t)rg * ld + cq * 4) = a;
}

hipError_t launch_colsum_groups(const float* X, const float* Y, int rows, int64_t ld, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(colsum_groups_kernel, dim3(epilogue_blocks(rows, ld)), dim3(256), 0, s, X, Y, rows, ld, out);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// parameter update (rbm.py:347-365): lambda shrink, EMA "speed", lagged apply
//   g   = (S / batch_size - weightcost * W0) / (1 + 2 lr l1 / (|W| + eps))
//   W'  = W * (1 - 2 lr l2) / (1 + 2 lr l1 / (|W| + eps)) + W_speed(old) * lr
//   Ws' = g + (W_speed - g) * momentum
// ----------------------------------------------------------------------------------
// NEWSPEED (with both halves): the parameters take the speed computed in this very pass -- phase 1 of
// step t-1 followed by phase 2 of step t of the overlapped data-parallel order, in one launch
template <bool DO_SPEED, bool DO_PARAMS, bool NEWSPEED = false>
__global__ __launch_bounds__(256) void update_kernel(float4* __restrict__ W, float4* __restrict__ Ws,
                                                     const float4* __restrict__ W0, const float4* __restrict__ S,
                                                     int64_t n4, float lr, float l1, float l2, float wc,
                                                     float mu, float inv_bs,
                                                     float* __restrict__ hb, float* __restrict__ hbs,
                                                     const float* __restrict__ s_h, int64_t H,
                                                     float* __restrict__ vb, float* __restrict__ vbs,
                                                     const float* __restrict__ s_v, int64_t V, float inv_rows,
                                                     const float* __restrict__ cost_sum, float cost_scale,
                                                     float* __restrict__ cost_out, int nslab, int64_t slab_stride4,
                                                     unsigned short* __restrict__ Wp)
{
    // nslab > 1 (whole rule only): S points at the split-K slabs of the statistics GEMM, summed here
    // in slab order exactly as sum_slabs_kernel would -- one launch and one pass over S fewer
    constexpr bool do_speed = DO_SPEED, do_params = DO_PARAMS;
    {   // biases, one element per thread of the leading blocks (multipliers are exactly 1, rbm.py:356)
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < H) {
            float sp = hbs[i];
            if (do_speed) {
                const float sn = upd_speed(upd_scale(s_h[i], inv_rows), sp, mu);
                hbs[i] = sn;
                if (NEWSPEED) sp = sn;
            }
            if (do_params) hb[i] = upd_param(hb[i], 1.0f, sp, lr);
        } else if (i < H + V) {
            const int64_t j = i - H;
            float sp = vbs[j];
            if (do_speed) {
                const float sn = upd_speed(upd_scale(s_v[j], inv_rows), sp, mu);
                vbs[j] = sn;
                if (NEWSPEED) sp = sn;
            }
            if (do_params) vb[j] = upd_param(vb[j], 1.0f, sp, lr);
        }
        if (i == 0 && cost_out && do_speed) cost_out[0] = cost_sum[0] * cost_scale;
    }
    const float two_lr_l1 = upd_two_lr_l1(lr, l1);
    const float decay = upd_decay(lr, l2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 sp = Ws[i];
        if constexpr (DO_SPEED && DO_PARAMS && NEWSPEED) {      // lambda_1 == 0; weight cost, if any, uses W0
            const float4 w = W[i], st = S[i];
            const float4 w0 = wc != 0.0f ? W0[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 sn, wn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float snj = upd_speed(upd_grad(comp(st, j), inv_bs, wc, comp(w0, j)), comp(sp, j), mu);
                setc(sn, j, snj);
                setc(wn, j, upd_param(comp(w, j), decay, snj, lr));
            }
            Ws[i] = sn;
            W[i] = wn;
            if (Wp) store_planes4(Wp, n4 * 4, i * 4, wn);
        } else if constexpr (DO_SPEED && DO_PARAMS) {
            const float4 w = W[i];
            float4 st;
            if (nslab <= 1) {
                st = S[i];
            } else {
                st = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s0 = 0; s0 < nslab; s0 += 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = s0 + u < nslab ? S[i + (int64_t)(s0 + u) * slab_stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { st.x += v[u].x; st.y += v[u].y; st.z += v[u].z; st.w += v[u].w; }
                }
            }
            const float4 wc0 = W0 ? W0[i] : w;
            float4 wn, sn;
            update_rule4(w, sp, st, wc0, inv_bs, wc, decay, l1, two_lr_l1, mu, lr, wn, sn);
            W[i] = wn;
            Ws[i] = sn;
            if (Wp) store_planes4(Wp, n4 * 4, i * 4, wn);
        } else if constexpr (DO_PARAMS) {           // lambda_1 == 0 (checked by the host): m = decay
            const float4 w = W[i];
            const float4 wn = make_float4(upd_param(w.x, decay, sp.x, lr), upd_param(w.y, decay, sp.y, lr),
                                          upd_param(w.z, decay, sp.z, lr), upd_param(w.w, decay, sp.w, lr));
            W[i] = wn;
            if (Wp) store_planes4(Wp, n4 * 4, i * 4, wn);
        } else {                                    // speeds only; weight cost, if any, uses W0
            const float4 st = S[i];
            const float4 w0 = wc != 0.0f ? W0[i] : make_float4(0.f, 0.f, 0.f, 0.f);     // host: W0 != NULL when wc != 0
            Ws[i] = make_float4(upd_speed(upd_grad(st.x, inv_bs, wc, w0.x), sp.x, mu),
                                upd_speed(upd_grad(st.y, inv_bs, wc, w0.y), sp.y, mu),
                                upd_speed(upd_grad(st.z, inv_bs, wc, w0.z), sp.z, mu),
                                upd_speed(upd_grad(st.w, inv_bs, wc, w0.w), sp.w, mu));
        }
    }
}

hipError_t launch_update(const mdbn_update_args& a, hipStream_t s, const float* slabs, int nslab,
                         int64_t slab_stride, unsigned short* Wp)
{
    const int64_t n4 = (a.V * a.ldh) >> 2;
    if (slabs && (a.phase != 0 || (slab_stride & 3))) return hipErrorInvalidValue;
    const float* S = slabs ? slabs : a.stats;
    if (!slabs) nslab = 1;
    const float* s_h = a.stats + a.V * a.ldh;
    const float* s_v = s_h + a.ldh;
    const int grid = (int)std::max<int64_t>(std::min<int64_t>((n4 + 255) / 256, 4096), (a.H + a.V + 255) / 256);
#define LAUNCH_UPDATE_(SP, PA, NS)                                                                     \
    hipLaunchKernelGGL((update_kernel<SP, PA, NS>), dim3(grid), dim3(256), 0, s, reinterpret_cast<float4*>(a.W), \
                       reinterpret_cast<float4*>(a.W_speed), reinterpret_cast<const float4*>(a.W0),    \
                       reinterpret_cast<const float4*>(S), n4, a.lr, a.lambda_1, a.lambda_2, a.weightcost, \
                       a.momentum, 1.0f / a.batch_size, a.hbias, a.hbias_speed, s_h, a.H, a.vbias,    \
                       a.vbias_speed, s_v, a.V, 1.0f / a.n_rows, s_v + a.ldv, a.cost_scale, a.cost_out, nslab, \
                       slab_stride >> 2, Wp)
#define LAUNCH_UPDATE(SP, PA) LAUNCH_UPDATE_(SP, PA, false)
#define LAUNCH_UPDATE3() LAUNCH_UPDATE_(true, true, true)
    if (a.phase == 1) LAUNCH_UPDATE(true, false);
    else if (a.phase == 2) LAUNCH_UPDATE(false, true);
    else if (a.phase == 3) LAUNCH_UPDATE3();
    else LAUNCH_UPDATE(true, true);
#undef LAUNCH_UPDATE
#undef LAUNCH_UPDATE3
#undef LAUNCH_UPDATE_
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// free energy (RBM rbm.py:166-171, GRBM rbm.py:684-688): one block per row;
// slabs hold the split-K partials of x W.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void free_energy_kernel(const float* __restrict__ slabs, int nsplit,
                                                          int64_t slab_stride, int64_t ldh, int H,
                                                          const float* __restrict__ hbias,
                                                          const float* __restrict__ x, int64_t ldv, int V,
                                                          const float* __restrict__ vbias, int gauss,
                                                          float* __restrict__ out)
{
    // F is a difference of two O(V) sums (cancellation): reduce the rows in f64 so that the
    // result carries only the GEMM's fp32 rounding, not the reduction's.
    __shared__ double red[4];
    const int64_t r = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        double a = (double)hbias[j];      // split-K partials summed in float64, rounded once
        for (int s = 0; s < nsplit; ++s) a += (double)slabs[(int64_t)s * slab_stride + r * ldh + j];
        acc -= (double)softplusf_((float)a);
    }
    for (int j = threadIdx.x; j < V; j += blockDim.x) {
        const float xv = x[r * ldv + j], b = vbias[j];
        if (gauss) { const float d = xv - b; acc += 0.5 * (double)d * (double)d; }
        else acc -= (double)xv * (double)b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[r] = (float)(red[0] + red[1] + red[2] + red[3]);
}

hipError_t launch_free_energy(const float* slabs, int nsplit, int64_t slab_stride, int64_t ldh, int H,
                              const float* hbias, const float* x, int64_t ldv, int V, const float* vbias,
                              int gauss, int64_t rows, float* out, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(free_energy_kernel, dim3((unsigned)rows), dim3(256), 0, s, slabs, nsplit, slab_stride,
                       ldh, H, hbias, x, ldv, V, vbias, gauss, out);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// the random matrices themselves
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rng_fill_kernel(float* __restrict__ out, int64_t rows, int64_t cols,
                                                       int64_t ld, PhiloxKey k, int normal)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rg = idx / cols, c = idx - rg * cols;
    const int64_t r0 = rg * 4;
    if (r0 >= rows) return;
    uint32_t wa[4], wb[4];
    philox_rows4(k, k.draw, k.row_offset + (uint64_t)r0, (uint32_t)c, wa);
    if (normal) philox_rows4(k, k.draw | MDBN_NORMAL_BIT, k.row_offset + (uint64_t)r0, (uint32_t)c, wb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r0 + r >= rows) break;
        float v = philox_u01(wa[r]);
        if (normal) v = sqrtf(-2.0f * logf(v)) * cosf(6.28318530717958647692f * philox_u01(wb[r]));
        out[(r0 + r) * ld + c] = v;
    }
}

hipError_t launch_rng_fill(float* out, int64_t rows, int64_t cols, int64_t ld, const PhiloxKey& k,
                           int normal, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const int64_t n = ((rows + 3) / 4) * cols;
    hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, rows, cols,
                       ld, k, normal);
    return hipGetLastError();
}


// ----------------------------------------------------------------------------------
// small elementwise / reduction kernels around the path (monitoring and host-side helpers that were
// torch expressions before): pseudo-likelihood pieces (rbm.py:421-447), reconstruction cost of given
// arrays (rbm.py:449-482, :690-699), tanh for HiddenLayer (mlp.py:103-107), finite check (the role of
// NanGuardMode, rbm.py:542-543).
// ----------------------------------------------------------------------------------
// xi = tensor.round(x) (round half away from zero); flip_col >= 0: that column becomes 1 - xi
__global__ __launch_bounds__(256) void round_flip_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                         int64_t cols, int64_t ld, int64_t flip_col)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld) return;
    const int64_t c = i % ld;
    float v = 0.f;
    if (c < cols) {
        const float a = x[i];
        v = copysignf(floorf(fabsf(a) + 0.5f), a);
        if (c == flip_col) v = 1.0f - v;
    }
    out[i] = v;
}

hipError_t launch_round_flip(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, int64_t flip_col, hipStream_t s)
{
    const int64_t n = rows * ld;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(round_flip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, rows, cols, ld, flip_col);
    return hipGetLastError();
}

// out[0] = -mean_r( n_visible * softplus(F(xi)_r - F(xi_flip)_r) )   (rbm.py:442); one block, fixed order
__global__ __launch_bounds__(256) void pl_cost_kernel(const float* __restrict__ fe, const float* __restrict__ fe_flip,
                                                      int64_t rows, float n_visible, float* __restrict__ out)
{
    __shared__ float red[8];
    float a = 0.f;
    for (int64_t r = threadIdx.x; r < rows; r += blockDim.x) a += softplusf_(fe[r] - fe_flip[r]);
    const float tot = block_sum(a, red);
    if (threadIdx.x == 0) out[0] = -(n_visible * tot) / (float)rows;
}

hipError_t launch_pl_cost(const float* fe, const float* fe_flip, int64_t rows, float n_visible, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(pl_cost_kernel, dim3(1), dim3(256), 0, s, fe, fe_flip, rows, n_visible, out);
    return hipGetLastError();
}

// reconstruction cost of given arrays: partial sums per block of BCE(sigmoid(pre), t) (gauss = 0) or
// (sigmoid(pre) - t)^2 (gauss = 1) over the live columns; summed by finalize (fixed order)
__global__ __launch_bounds__(256) void recon_cost_kernel(const float* __restrict__ pre, int64_t ldp, const float* __restrict__ tgt,
                                                         int64_t ldt, int64_t rows, int64_t cols, int gauss,
                                                         float* __restrict__ partials)
{
    __shared__ float red[8];
    float a = 0.f;
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        const float x = pre[r * ldp + c], t = tgt[r * ldt + c];
        if (gauss) { const float d = sigmoidf_(x) - t; a += d * d; }
        else a += t * softplusf_(-x) + (1.0f - t) * softplusf_(x);
    }
    const float tot = block_sum(a, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(64) void sum_partials_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out)
{
    float a = 0.f;
    for (int k = threadIdx.x; k < n; k += 64) a += partials[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (threadIdx.x == 0) out[0] = a * scale;
}

hipError_t launch_recon_cost(const float* pre, int64_t ldp, const float* tgt, int64_t ldt, int64_t rows, int64_t cols, int gauss,
                             float scale, float* partials, int n_partials, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(recon_cost_kernel, dim3(n_partials), dim3(256), 0, s, pre, ldp, tgt, ldt, rows, cols, gauss, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, partials, n_partials, scale, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void tanh_kernel(float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ld) return;
    if (i % ld < cols) x[i] = tanhf(x[i]);
}

hipError_t launch_tanh(float* x, int64_t rows, int64_t cols, int64_t ld, hipStream_t s)
{
    const int64_t n = rows * ld;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tanh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, rows, cols, ld);
    return hipGetLastError();
}

// bfloat16 wire format of the data-parallel statistics (opt-in reporting mode, mdbn_amd/dist.py): float32 -> bfloat16 with
// round-to-nearest-even (a plain cast: v_cvt_pk_bf16_f32, NaN stays NaN) and back (exact).  16 bytes per thread and access.
__global__ __launch_bounds__(256) void narrow_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n)
{
    const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(x + i8), b = *reinterpret_cast<const float4*>(x + i8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 lo = (__bf16)v[2 * e], hi = (__bf16)v[2 * e + 1];
            w[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
        }
        *reinterpret_cast<uint4*>(y + i8) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int64_t i = i8; i < n; ++i) { const __bf16 t = (__bf16)x[i]; y[i] = __builtin_bit_cast(unsigned short, t); }
    }
}

__global__ __launch_bounds__(256) void widen_bf16_kernel(const unsigned short* __restrict__ y, float* __restrict__ x, int64_t n)
{
    const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 + 8 <= n) {
        const uint4 w = *reinterpret_cast<const uint4*>(y + i8);
        const unsigned u[4] = {w.x, w.y, w.z, w.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, u[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, u[e] & 0xffff0000u);
        }
        *reinterpret_cast<float4*>(x + i8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(x + i8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        for (int64_t i = i8; i < n; ++i) x[i] = __builtin_bit_cast(float, (unsigned)y[i] << 16);
    }
}

hipError_t launch_narrow_bf16(const float* x, unsigned short* y, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

hipError_t launch_widen_bf16(const unsigned short* y, float* x, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, y, x, n);
    return hipGetLastError();
}

// count[0] += number of NaN / Inf entries of x[0..n)
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float* __restrict__ x, int64_t n, int* __restrict__ count)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad += (__builtin_bit_cast(unsigned, x[i]) & 0x7f800000u) == 0x7f800000u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

hipError_t launch_count_nonfinite(const float* x, int64_t n, int* count, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3(grid), dim3(256), 0, s, x, n, count);
    return hipGetLastError();
}

}  // namespace mdbn
