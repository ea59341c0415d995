// GEMMs of the CD step on PRE-SPLIT bf16 planes (gfx950 / MI355X).
//
// gemm_bf16x6_kernel (mdbn_kernels.hip) keeps every operand f32 in HBM and splits it into three bf16
// pieces inside the kernel, in every block that loads it: its producer waves spend their time on split
// arithmetic and LDS stores.  Here every tensor of the step is split ONCE, where it is produced (the gather,
// the activation epilogues, the weight update), into three bf16 planes x = p1 + p2 + p3 (exact), and the
// GEMM only COPIES plane tiles into LDS with LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU, no
// ds_write) and issues the six piece products (three when one operand holds 0/1 samples) on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation -- the same products in the same order as
// gemm_bf16x6_kernel, so an unsplit GEMM gives the same bits.
//
// Operand layouts (planes are [3][rows][ld] bf16 in HBM, natural row-major like their f32 originals):
//   ROW operand: the reduction index k is contiguous in memory ([rows][k]).  LDS image [128 rows][32 k]
//       (64-byte rows), 16-byte chunk c of row r stored at chunk c ^ ((r >> 2) & 3): the MFMA fragment
//       (8 consecutive k of one row) is ONE ds_read_b128, conflict-free.
//   COL operand: k is the ROW index in memory ([k][rows]), i.e. the operand is used transposed
//       (x^T, ph in the statistics GEMM; W in propup).  LDS image [32 k][128 rows] (256-byte rows), chunk c of
//       k-row k stored at c ^ ((k & 3) << 2); the fragment is read with two ds_read_b64_tr_b16, gfx950's
//       transposing LDS read -- no transposed copy of any tensor exists anywhere.
//   LDS-DMA writes linearly (wave base + lane * 16), so the swizzles are applied to the per-lane SOURCE
//   address (cdna_hip_programming.md rule 21).
// Block: 128x128 tile, 32-deep stages, 3-stage LDS ring (144 KB), 4 MFMA waves (one per SIMD, 64x64 each)
// + 4 loader waves.  A loader waits for stage it+1 (counted vmcnt), joins the barrier that ends stage it,
// and only then issues stage it+3 into the slot just freed: issue latency stays off the barrier's path
// and two stages are always in flight.  Measured (scripts/experiments/planes_gemm.py): ~1900 cycles per
// stage against 1536 of MFMA issue; the chip holds only 1.3-1.6 GHz under this load, which is what bounds
// the wall time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "mdbn_kernels.h"
#include "mdbn_device.h"

namespace mdbn {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef short ps16x4 __attribute__((ext_vector_type(4)));
typedef short ps16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));

#ifndef PL_ARGS_EARLY
#define PL_ARGS_EARLY 1
#endif
constexpr int PL_PLANE = 8192, PL_STAGE = 6 * PL_PLANE, PL_NSTAGE = 3, PL_LW = 4;

__device__ __forceinline__ void pl_glds16(const void* g, unsigned lds_off, char* smem)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
}

// LDS-image swizzles, per MFMA shape MS (32: v_mfma_f32_32x32x16_bf16, 16: v_mfma_f32_16x16x32_bf16), chosen so that the
// fragment reads of that shape are bank-conflict free (derivations: DESIGN.md section 3):
//   ROW image [128 rows][64 B]: 16-byte chunk c of row r lives at chunk c ^ pl_row_swz(r)
//   COL image [32 k][256 B]   : 16-byte chunk c of k-row k lives at chunk c ^ pl_col_swz(k)
template <int MS> __device__ __forceinline__ int pl_row_swz(int row)
{
    return MS == 32 ? ((row >> 2) & 3) : ((4 - ((row >> 2) & 3)) & 3);
}
template <int MS> __device__ __forceinline__ int pl_col_swz(int k)
{
    return MS == 32 ? ((k & 3) << 2) : (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
}

// One loader wave's share of the staging: instructions q = w, w + LW, ... of the 8 * (AP + 3) per stage
// (8 per plane: 1 KiB each).
// diagnostic builds (-DMDBN_STAMP): wall-clock (100 MHz) stamps of one propup workgroup's phases, slot per phase
#ifdef MDBN_STAMP
// which launch is stamped: propup (default) or, with -DMDBN_STAMP_NARROW, propdown on narrow tiles
#if defined(MDBN_STAMP_NARROW)
#define PL_STAMP_COND (LA == LAY_K && LB == LAY_K && AP == 1 && BN == 64)
#elif defined(MDBN_STAMP_STATS)
#define PL_STAMP_COND (LA == LAY_MN && LB == LAY_MN && AP == 3)
#else
#define PL_STAMP_COND (LA == LAY_K && LB == LAY_MN && AP == 3)
#endif
#define PL_STAMP(SLOT)                                                                        \
    do {                                                                                      \
        if (PL_STAMP_COND && g.stamps && w == 0 && lane == 0)                                 \
            g.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64();                     \
    } while (0)
#else
#define PL_STAMP(SLOT) do {} while (0)
#endif
// EARLYW (statistics GEMMs): between their LDS-DMA issues the loader waves also do memory-bound work that does not depend
// on this GEMM -- UpdEpi.early = 1: the PARAMETER half of the fused update of this workgroup's W tile (+ the gather of the
// next minibatch, GatherAhead); UpdEpi.early = 2 (data-parallel): the whole deferred update of the previous step.  One item
// per stage, loads one stage ahead: see the comment at the phases below.
struct EarlySpeed { float4 sp[16], w0[16]; };      // a loader lane's share of the tile's old speed (+ frozen W0), rows 8 j + lt / 32
// BN = 128 | 64: columns of the output tile.  BN = 64 (ROW B operand only: propdown) halves the B image -- four 16-row
// instructions per plane instead of eight -- so that an output with 256 tiles of 128 x 64 needs no split-K.
// bias / cost half of the previous step's deferred update (data-parallel order), by the four MFMA waves of every workgroup
// ahead of their main loop: the same arithmetic as update_kernel<true, true, true>'s leading blocks
__device__ __forceinline__ void deferred_bias_update(const DeferredBias& d, int wave, int lane)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + wave * 64 + lane; i < d.H + d.V; i += (int64_t)gridDim.x * 256) {
        if (i < d.H) {
            const float sn = upd_speed(upd_scale(d.s_h[i], d.inv_rows), d.hbs[i], d.mu);
            d.hbs[i] = sn;
            d.hb[i] = upd_param(d.hb[i], 1.0f, sn, d.lr);
        } else {
            const int64_t j = i - d.H;
            const float sn = upd_speed(upd_scale(d.s_v[j], d.inv_rows), d.vbs[j], d.mu);
            d.vbs[j] = sn;
            d.vb[j] = upd_param(d.vb[j], 1.0f, sn, d.lr);
        }
    }
    if (blockIdx.x == 0 && wave == 0 && lane == 0 && d.cost_out) d.cost_out[0] = d.cost_sum[0] * d.cost_scale;
}

template <int LA, int LB, int AP, int MS, bool EARLYW = false, int BN = 128>
__device__ __forceinline__ void pl_loader(const PlaneGemmArgs& g, char* smem, int w, int lane, int m0, int n0, int kbeg, int nt,
                                          EarlySpeed* es = nullptr)
{
    PL_STAMP(0);
    static_assert(BN == 128 || (BN == 64 && LB == LAY_K), "a 64-column tile is implemented for a ROW B operand");
    // AP = planes of A (3 | 1); AP = 0 is the bf16-input REPORTING mode: one plane of each operand, one product
    constexpr int NA = AP == 3 ? 3 : 1, NB = AP == 0 ? 1 : 3;
    constexpr int QB = BN / 16;                      // 1-KiB instructions per B plane
    constexpr int NQ = 8 * NA + QB * NB, PER = NQ / PL_LW;
    static_assert(NQ % PL_LW == 0, "staging instructions must divide over the loader waves");
    const char* src[PER];
    unsigned dst[PER];
    int64_t step[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = w + j * PL_LW;
        const bool isA = q < 8 * NA;
        const int plane = isA ? q / 8 : (q - 8 * NA) / QB, sub = isA ? (q & 7) : (q - 8 * NA) % QB;
        const int lay = isA ? LA : LB;
        const unsigned short* base = isA ? g.A + plane * g.pa : g.B + plane * g.pb;
        const int64_t ld = isA ? g.lda : g.ldb;
        const int mn0 = isA ? m0 : n0;
        dst[j] = (isA ? plane : 3 + plane) * PL_PLANE + sub * 1024;
        if (lay == LAY_K) {      // ROW: 16 rows x 64 B per instruction
            const int row = 16 * sub + (lane >> 2);
            const int c = (lane & 3) ^ pl_row_swz<MS>(row);
            src[j] = reinterpret_cast<const char*>(base + (int64_t)(mn0 + row) * ld + kbeg + 8 * c);
            step[j] = 64;
        } else {                 // COL: 4 k-rows x 256 B per instruction
            const int k = 4 * sub + (lane >> 4);
            const int ch = (lane & 15) ^ pl_col_swz<MS>(k);
            src[j] = reinterpret_cast<const char*>(base + (int64_t)(kbeg + k) * ld + mn0 + 8 * ch);
            step[j] = 64 * ld;
        }
    }
#define PL_ISSUE(T)                                                                           \
    do {                                                                                      \
        const unsigned so = ((T) % PL_NSTAGE) * PL_STAGE;                                     \
        _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                     \
            pl_glds16(src[j], so + dst[j], smem);                                             \
            src[j] += step[j];                                                                \
        }                                                                                     \
    } while (0)
    // (Rejected on the way, same-box builds, profiles/r02z_*_variants.log: a loader that copies through registers
    // -- global_load_dwordx4 -> ds_write_b128, same images -- 159.4 vs 150.9 us per step; partial tiles stored in whole
    // rows through the LDS-parked tile 153.7 vs 153.0 / 153.5.)
    // Ramp-up: LDS-DMA issue BLOCKS at the rate the path moves data (~58 GB/s per CU: the three stages of the ring take
    // 2.5 us to issue, scripts/experiments/planes_stamps.py), so a loader that issued all three before waiting for the
    // first kept the MFMA waves idle for 2.8 us per launch.  Stage 0 alone goes first; the others follow once the MFMA
    // waves are running on it.
    PL_STAMP(1);
    PL_ISSUE(0);
    PL_STAMP(2);
    // gather-ahead: the source rows of this workgroup's <= 4 rows of the NEXT minibatch, resolved HERE -- four index loads
    // issued together under the wait for stage 0, which the loader sits out anyway.  (Resolved where the gather phase
    // begins, one `epi`-style lookup per row, they were four memory round trips in a row in the middle of the main loop,
    // each behind a `s_waitcnt vmcnt(0)` that also drained the LDS-DMA ring: found by scanning the ISA for loads waited
    // for at once; same-box A/B 140.0 -> 139.0 us per step, profiles/r04zv_ga_idx_ab.log.)
    int64_t ga_srow[4] = {0, 0, 0, 0};
    if constexpr (EARLYW) {
        if (g.upd.early && g.ga.idx) {
            int rr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (int)blockIdx.x * g.ga.rpw + (i < g.ga.rpw ? i : 0);
                rr[i] = r < g.ga.B ? r : g.ga.B - 1;
            }
            if (g.ga.idx64) {
                const int64_t* ix = reinterpret_cast<const int64_t*>(g.ga.idx);
#pragma unroll
                for (int i = 0; i < 4; ++i) ga_srow[i] = ix[rr[i]];
            } else {
                const int32_t* ix = reinterpret_cast<const int32_t*>(g.ga.idx);
#pragma unroll
                for (int i = 0; i < 4; ++i) ga_srow[i] = (int64_t)ix[rr[i]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t t = ga_srow[i];
                if (t < 0) t += g.ga.n_rows;
                ga_srow[i] = t < 0 ? 0 : (t >= g.ga.n_rows ? g.ga.n_rows - 1 : t);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PL_STAMP(3);
    __builtin_amdgcn_s_barrier();                    // stage 0 landed
    if (nt > 1) { PL_ISSUE(1); }
    if (nt > 2) { PL_ISSUE(2); }
    // EARLYW side work, one ITEM per stage: first the chunk items of the W tile (CPI 8-row chunks each, 16 / CPI items),
    // then the gather-ahead units (one 256-octet pass of one row of the next minibatch each).  An item's loads are issued
    // ONE STAGE BEFORE it is applied, ahead of that stage's DMAs, so a whole MFMA stage (~1.1 us) hides their HBM latency;
    // they are applied after the DMAs of the following stage.  hipcc cannot express that: with loads, stores and LDS-DMA
    // pending together its waitcnt insertion falls back to vmcnt(0) before the first use (seen in the ISA: every apply then
    // waited for the loads just issued and for the stage's DMAs).  So the item loads are inline-asm loads hipcc does not
    // count, each consumed behind a counted wait statement that names its destination registers "+v"
    // (cdna_hip_programming.md 5.7, form (ii)): vmcnt(PER + younger item loads) = this stage's DMAs and the next item's
    // loads may fly, everything older -- the item's own loads, issued a stage ago -- has landed (VMEM operations retire in
    // issue order).  The phases are straight-line code (fully unrolled, loads unconditional at valid addresses), so no
    // register that a load is still writing is ever copied or merged; its stores are compiler-visible and younger than the
    // stage they follow, so the counted waits of the DMA ring stay valid.  The host sizes the items to fit the stages that
    // issue a DMA.
    const int lt = w * 64 + lane;
    int it = 0;
#define LD_SYNC()                                                                             \
    do {                                                                                      \
        if (it + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");          \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
        __builtin_amdgcn_s_barrier();                /* every read of stage `it` is done: its slot is free */ \
    } while (0)
#define ASM_LOAD4(DST, PTR) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(DST) : "v"(PTR) : "memory")
// the counted wait ahead of an item's first consumer: a wait-only statement + a scheduling barrier (5.7, form (iii)).
// (Form (ii), naming the destinations "+v", made hipcc COPY them into fresh registers ahead of the wait on one path --
// reading registers whose load had not landed; found by the ISA audit, scripts/experiments/audit_asm_loads.py.)
#define ASM_WAIT(N) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    if constexpr (EARLYW) {
        if (g.upd.early == 2) {
            // the whole deferred update of the previous step (phase 3), 16 chunk items, one per stage (host: nt >= 20):
            // speed' = g + (speed - g) mu with g = Sprev / batch_size - wc W0, W' = W decay + speed' lr -- update_kernel's
            // NEWSPEED branch on the same operands.  Four loads per item (W0 from a dummy address when wc == 0).
            const float decay = upd_decay(g.upd.lr, g.upd.l2);
            const int64_t lane_off = (int64_t)(m0 + (lt >> 5)) * g.upd.ld + n0 + 4 * (lt & 31);
            const bool has_wc = g.upd.wc != 0.0f;
            const float* w0base = has_wc ? g.upd.W0 : g.upd.Ws;
            pf32x4 dw[16], ds[16], dt[16], d0[16];
            ASM_LOAD4(dw[0], g.upd.W + lane_off); ASM_LOAD4(ds[0], g.upd.Ws + lane_off);
            ASM_LOAD4(dt[0], g.upd.Sprev + lane_off); ASM_LOAD4(d0[0], w0base + lane_off);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                LD_SYNC();
                if (i + 1 < 16) {
                    const int64_t o1 = lane_off + (int64_t)8 * (i + 1) * g.upd.ld;
                    ASM_LOAD4(dw[i + 1], g.upd.W + o1); ASM_LOAD4(ds[i + 1], g.upd.Ws + o1);
                    ASM_LOAD4(dt[i + 1], g.upd.Sprev + o1); ASM_LOAD4(d0[i + 1], w0base + o1);
                }
                PL_ISSUE(it + 3);
                if (i + 1 < 16) ASM_WAIT(PER + 4); else ASM_WAIT(PER);
                const int64_t off = lane_off + (int64_t)8 * i * g.upd.ld;
                float4 sn, wn;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float w0j = has_wc ? d0[i][j] : 0.0f;
                    const float snj = upd_speed(upd_grad(dt[i][j], g.upd.inv_bs, g.upd.wc, w0j), ds[i][j], g.upd.mu);
                    setc(sn, j, snj);
                    setc(wn, j, upd_param(dw[i][j], decay, snj, g.upd.lr));
                }
                *reinterpret_cast<float4*>(g.upd.Ws + off) = sn;
                *reinterpret_cast<float4*>(g.upd.W + off) = wn;
                if (g.upd.Wp) store_planes4(g.upd.Wp, g.upd.wp_stride, off, wn);
                ++it;
            }
        } else if (g.upd.early) {
            const float decay = upd_decay(g.upd.lr, g.upd.l2);
            const int64_t lane_off = (int64_t)(m0 + (lt >> 5)) * g.upd.ld + n0 + 4 * (lt & 31);
#define CW_STORE(WV, SV, CH)                                                                  \
    do {                                                                                      \
        const int64_t off_ = lane_off + (int64_t)8 * (CH) * g.upd.ld;                         \
        const float4 w4_ = make_float4(WV[0], WV[1], WV[2], WV[3]), s4_ = make_float4(SV[0], SV[1], SV[2], SV[3]); \
        float4 wn_;                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) setc(wn_, j_, upd_param(comp(w4_, j_), decay, comp(s4_, j_), g.upd.lr)); \
        *reinterpret_cast<float4*>(g.upd.W + off_) = wn_;                                     \
        if (g.upd.Wp) store_planes4(g.upd.Wp, g.upd.wp_stride, off_, wn_);                    \
    } while (0)
            if (nt >= 20) {                          // one chunk per stage: 16 items
                pf32x4 cw[17], cs[17];
                ASM_LOAD4(cw[0], g.upd.W + lane_off); ASM_LOAD4(cs[0], g.upd.Ws + lane_off);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    LD_SYNC();
                    if (i + 1 < 16) {
                        ASM_LOAD4(cw[i + 1], g.upd.W + lane_off + (int64_t)8 * (i + 1) * g.upd.ld);
                        ASM_LOAD4(cs[i + 1], g.upd.Ws + lane_off + (int64_t)8 * (i + 1) * g.upd.ld);
                    }
                    PL_ISSUE(it + 3);
                    if (i + 1 < 16) ASM_WAIT(PER + 2); else ASM_WAIT(PER);
                    CW_STORE(cw[i], cs[i], i);
                    es->sp[i] = make_float4(cs[i][0], cs[i][1], cs[i][2], cs[i][3]);      // kept for the speed epilogue
                    ++it;
                }
            } else {                                 // two chunks per stage: 8 items (host: nt >= 12)
                pf32x4 cw[18], cs[18];
                ASM_LOAD4(cw[0], g.upd.W + lane_off); ASM_LOAD4(cs[0], g.upd.Ws + lane_off);
                ASM_LOAD4(cw[1], g.upd.W + lane_off + (int64_t)8 * g.upd.ld); ASM_LOAD4(cs[1], g.upd.Ws + lane_off + (int64_t)8 * g.upd.ld);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    LD_SYNC();
                    if (i + 1 < 8) {
                        ASM_LOAD4(cw[2 * i + 2], g.upd.W + lane_off + (int64_t)8 * (2 * i + 2) * g.upd.ld);
                        ASM_LOAD4(cs[2 * i + 2], g.upd.Ws + lane_off + (int64_t)8 * (2 * i + 2) * g.upd.ld);
                        ASM_LOAD4(cw[2 * i + 3], g.upd.W + lane_off + (int64_t)8 * (2 * i + 3) * g.upd.ld);
                        ASM_LOAD4(cs[2 * i + 3], g.upd.Ws + lane_off + (int64_t)8 * (2 * i + 3) * g.upd.ld);
                    }
                    PL_ISSUE(it + 3);
                    if (i + 1 < 8) ASM_WAIT(PER + 4); else ASM_WAIT(PER);
                    CW_STORE(cw[2 * i], cs[2 * i], 2 * i);
                    CW_STORE(cw[2 * i + 1], cs[2 * i + 1], 2 * i + 1);
                    es->sp[2 * i] = make_float4(cs[2 * i][0], cs[2 * i][1], cs[2 * i][2], cs[2 * i][3]);
                    es->sp[2 * i + 1] = make_float4(cs[2 * i + 1][0], cs[2 * i + 1][1], cs[2 * i + 1][2], cs[2 * i + 1][3]);
                    ++it;
                }
            }
#undef CW_STORE
        }
        if (g.upd.early) {
            if (g.ga.idx) {
                // gather-ahead: rows blockIdx * rpw .. of the next minibatch, <= 4 units (host); source rows resolved once
                // (as gather_planes_kernel).  The loads are unconditional at clamped, valid addresses: only stores are predicated.
                const int64_t (&srow)[4] = ga_srow;         // (resolved at the head of the loader, under the wait for stage 0)
                const int64_t ld8 = g.ga.ld >> 3;
                const int nunits = g.ga.rpw * g.ga.passes;
                pf32x4 ga[4], gb[4];
#define GU_ADDR(U)                                                                            \
    const int i_r0 = (U) / g.ga.passes, pass_ = (U) - i_r0 * g.ga.passes, i_r = i_r0 < 4 ? i_r0 : 3;   \
    const int64_t c_ = (int64_t)pass_ * 256 + lt, cc_ = c_ < ld8 ? c_ : ld8 - 1;              \
    const int64_t sr_ = i_r == 0 ? srow[0] : i_r == 1 ? srow[1] : i_r == 2 ? srow[2] : srow[3]; \
    const float* p_ = g.ga.src + sr_ * g.ga.ld_src + 8 * cc_;
#define GU_LOAD(K, U) do { GU_ADDR(U) ASM_LOAD4(ga[K], p_); ASM_LOAD4(gb[K], p_ + 4); } while (0)
#define GU_APPLY(K, U)                                                                        \
    do {                                                                                      \
        const int i_r = (U) / g.ga.passes, pass_ = (U) - i_r * g.ga.passes;                   \
        const int r_ = (int)blockIdx.x * g.ga.rpw + i_r;                                      \
        const int64_t c_ = (int64_t)pass_ * 256 + lt;                                         \
        if ((U) < nunits && r_ < g.ga.B && c_ < ld8) {                                        \
            const float v_[8] = {ga[K][0], ga[K][1], ga[K][2], ga[K][3], gb[K][0], gb[K][1], gb[K][2], gb[K][3]}; \
            unsigned short q_[3][8];                                                          \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) split3(v_[j_], q_[0][j_], q_[1][j_], q_[2][j_]); \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                \
                uint4 wv_;                                                                    \
                wv_.x = q_[p_][0] | ((unsigned)q_[p_][1] << 16); wv_.y = q_[p_][2] | ((unsigned)q_[p_][3] << 16); \
                wv_.z = q_[p_][4] | ((unsigned)q_[p_][5] << 16); wv_.w = q_[p_][6] | ((unsigned)q_[p_][7] << 16); \
                *reinterpret_cast<uint4*>(g.ga.P + p_ * g.ga.plane_stride + (int64_t)r_ * g.ga.ld + 8 * c_) = wv_; \
            }                                                                                 \
        }                                                                                     \
    } while (0)
                // Four unit slots, always: stage 0 of the phase only issues unit 0's loads; stage k + 1 issues unit k + 1's
                // and applies unit k.  A slot beyond nunits loads from a clamped, valid address and stores nothing, so every
                // load is consumed and every counted wait is UNCONDITIONAL straight-line code -- which is what lets the ISA
                // audit (mdbn_amd/isa_audit.py, run by the build) prove that no destination register is touched before its
                // wait on any path.  (The first version nested the slots in run-time conditions; hipcc compiled the two
                // complementary waits of a slot as two independently guarded instructions, and a surplus load whose result
                // was never consumed had its destination reused as an address register: memory fault.)
                LD_SYNC(); GU_LOAD(0, 0); PL_ISSUE(it + 3); ++it;
#define GU_SLOT(K)                                                                            \
    do {                                                                                      \
        LD_SYNC();                                                                            \
        if ((K) < 3) { GU_LOAD((K) + 1, (K) + 1); }                                           \
        PL_ISSUE(it + 3);                                                                     \
        if ((K) < 3) ASM_WAIT(PER + 2); else ASM_WAIT(PER);                                   \
        GU_APPLY(K, K); ++it;                                                                 \
    } while (0)
                GU_SLOT(0); GU_SLOT(1); GU_SLOT(2); GU_SLOT(3);
#undef GU_SLOT
#undef GU_LOAD
#undef GU_APPLY
#undef GU_ADDR
            }
        }
    }
    for (; it < nt; ++it) {
        // stage it + 1 must have landed before the MFMA waves pass barrier `it`; stage it + 2 may still fly
        LD_SYNC();
        if (it + 3 < nt) { PL_ISSUE(it + 3); }
    }
#undef ASM_LOAD4
#undef ASM_WAIT
#undef LD_SYNC
    if constexpr (EARLYW) {
        if (g.upd.early == 1) {
            // the old speed of this lane's 16 chunks is still in registers (es->sp, from the chunk phase); the MFMA waves are
            // on the last stages: fetch the frozen W0 (when there is one) for the speed epilogue now, so that only its
            // stores follow the main loop
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t off = (int64_t)(m0 + 8 * j + (lt >> 5)) * g.upd.ld + n0 + 4 * (lt & 31);
                es->w0[j] = g.upd.W0 ? *reinterpret_cast<const float4*>(g.upd.W0 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    PL_STAMP(4);
#undef PL_ISSUE
}

template <int LAY>
__device__ __forceinline__ pbf16x8 pl_frag(const char* plane, int off0, int off1)
{
    if constexpr (LAY == LAY_K) {
        return *reinterpret_cast<const pbf16x8*>(plane + off0);
    } else {
        const ps16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(plane + off0));
        const ps16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(plane + off1));
        const ps16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(pbf16x8, v);
    }
}

// MFMA waves on v_mfma_f32_16x16x32_bf16: one MFMA spans the whole 32-deep stage, a wave owns 4 x 4 tiles of 16 x 16.
// The chip is power-limited under this load (1.3-1.6 GHz with the 32x32x16 shape) and holds a higher clock on this shape
// at equal cycles per FLOP (1.7 GHz: scripts/experiments/planes_gemm.py; MI355X_MICROARCH.md, DVFS give-back item 7).
// Four fragment register groups, one per operand half (two 16-row blocks x planes, 96 VGPRs in all).  The stage's tiles
// are visited quarter by quarter in a serpentine -- (lo,lo) (lo,hi) | (hi,hi) (hi,lo), next stage (lo,hi) (lo,lo) |
// (hi,lo) (hi,hi) -- so that consecutive quarters share one half and every quarter prefetches exactly ONE half under its
// MFMAs: this stage's in the first two quarters, the NEXT stage's after the mid-stage barrier `|`.
typedef float pf32x4a __attribute__((ext_vector_type(4)));

// fragment byte offsets inside a plane image for the 16x16x32 shape: [16-row block][first / second tr read]
template <int LA, int LB>
__device__ __forceinline__ void pl_offsets16(int lane, int wm, int wn, int (&offA)[4][2], int (&offB)[4][2])
{
    const int c16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        {
            const int row = wm + 16 * b + c16;
            if (LA == LAY_K) { offA[b][0] = row * 64 + ((q ^ pl_row_swz<16>(row)) << 4); offA[b][1] = 0; }
            else {
                const int ch = (wm + 16 * b) / 8 + ((c16 & 3) >> 1);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int k = 8 * q + 4 * tt + (c16 >> 2);
                    offA[b][tt] = k * 256 + ((ch ^ pl_col_swz<16>(k)) << 4) + 8 * (c16 & 1);
                }
            }
        }
        {
            const int row = wn + 16 * b + c16;
            if (LB == LAY_K) { offB[b][0] = row * 64 + ((q ^ pl_row_swz<16>(row)) << 4); offB[b][1] = 0; }
            else {
                const int ch = (wn + 16 * b) / 8 + ((c16 & 3) >> 1);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int k = 8 * q + 4 * tt + (c16 >> 2);
                    offB[b][tt] = k * 256 + ((ch ^ pl_col_swz<16>(k)) << 4) + 8 * (c16 & 1);
                }
            }
        }
    }
}

// Building blocks of the 16x16x32 consumers (pl_consume16, pl_consume16_bal); they expect offA / offB / acc and the
// constants NA, NB, RA, RB, NM in scope.
#define RD_A(FA, BASE, HALF)                                                                  \
    _Pragma("unroll") for (int pl = 0; pl < NA; ++pl)                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
            FA[pl][i] = pl_frag<LA>((BASE) + pl * PL_PLANE, offA[2 * (HALF) + i][0], offA[2 * (HALF) + i][1]);
#define RD_B(FB, BASE, HALF)                                                                  \
    _Pragma("unroll") for (int pl = 0; pl < NB; ++pl)                                         \
        _Pragma("unroll") for (int i = 0; i < NBH; ++i)                                       \
            FB[pl][i] = pl_frag<LB>((BASE) + (3 + pl) * PL_PLANE, offB[NBH * (HALF) + i][0], offB[NBH * (HALF) + i][1]);
// PL_STAGE_ACC = 1 (default): the products of one 32-deep stage are chained into a STAGE-LOCAL accumulator that starts at
// zero (smallest products first), and the stage's sum is added to the tile's accumulator with one v_add_f32 per element.
// A tile's accumulator then takes ONE rounding at its own (large) magnitude per stage instead of six (three): the chained
// roundings happen at the magnitude of a 32-term partial sum.  At K = 4096 the probabilities' worst error against float64
// drops from 2.2e-6 to ~1e-6 (SURVEY 8d asks 2e-6); measured cost: see docs/LABNOTES.md "Stage-local accumulators".
#ifndef PL_STAGE_ACC
#define PL_STAGE_ACC 1
#endif
#define MMQ_CHAIN(T, FA, FB, a, b, C0)                                                        \
    do {                                                                                      \
        T = (C0);                                                                             \
        if constexpr (AP == 3) {              /* smallest products first */                   \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[2][a], FB[0][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1][a], FB[0][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);      \
        } else if constexpr (AP == 1) {   /* A = a1 exactly: a1 b3 + a1 b2 + a1 b1 is the full product */ \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);      \
        } else {        /* bf16 inputs: the leading pieces only */                            \
            T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);      \
        }                                                                                     \
    } while (0)
#if PL_STAGE_ACC
// (propup layout only: LA == LAY_K && LB == LAY_MN)
// tile i's stage sum is added while tile i + 1's chain occupies the matrix pipe (the add would otherwise wait for the
// last MFMA of its own chain with nothing to issue)
#define MMQ(FA, FB, AH, BH)                                                                   \
    do {                                                                                      \
        if constexpr (AP == 0 || !(LA == LAY_K && LB == LAY_MN)) {                            \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                                     \
                _Pragma("unroll") for (int b = 0; b < NBH; ++b) {                             \
                    pf32x4a& c = acc[2 * (AH) + a][NBH * (BH) + b];                           \
                    MMQ_CHAIN(c, FA, FB, a, b, c);                                            \
                }                                                                             \
        } else {                                                                              \
            pf32x4a tq[2 * NBH];                                                              \
            _Pragma("unroll") for (int i = 0; i <= 2 * NBH; ++i) {                            \
                if (i < 2 * NBH) MMQ_CHAIN(tq[i], FA, FB, i / NBH, i % NBH, (pf32x4a{0.f, 0.f, 0.f, 0.f})); \
                if (i > 0) acc[2 * (AH) + (i - 1) / NBH][NBH * (BH) + (i - 1) % NBH] += tq[i - 1]; \
            }                                                                                 \
        }                                                                                     \
    } while (0)
#else
#define MMQ(FA, FB, AH, BH)                                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                             \
        _Pragma("unroll") for (int b = 0; b < NBH; ++b) {                                     \
            pf32x4a& c = acc[2 * (AH) + a][NBH * (BH) + b];                                   \
            MMQ_CHAIN(c, FA, FB, a, b, c);                                                    \
        }
#endif
// issue order: the prefetched half's LDS reads one by one under this quarter's MFMAs
#define ORD(NR)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < ((NR) < NM ? (NR) : NM); ++i_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
    }                                                                                         \
    if ((NR) > NM) __builtin_amdgcn_sched_group_barrier(0x100, (NR) - NM, 0);                 \
    if (NM > (NR)) __builtin_amdgcn_sched_group_barrier(0x008, NM - (NR), 0);
// every LDS read of a stage must have RETURNED before the barrier that frees its slot: pin the halves read in the
// first two quarters ahead of the barrier (hipcc may sink loads past s_barrier, DESIGN.md "guarded reload")
#define PIN(FA, NPL)                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < (NPL); ++pl)                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(FA[pl][i]));
#define PINB(FB, NPL)                                                                         \
    _Pragma("unroll") for (int pl = 0; pl < (NPL); ++pl)                                      \
        _Pragma("unroll") for (int i = 0; i < NBH; ++i) asm volatile("" :: "v"(FB[pl][i]));
// The two stage bodies of the serpentine.  EVEN enters holding (Alo, Blo) of its stage and leaves holding (Alo, Bhi) of
// the next; ODD enters holding (Alo, Bhi) and leaves holding (Alo, Blo).
#define PL_STAGE_EVEN(BASE, NEXT, BARRIER)                                                    \
    do {                                                                                      \
        RD_B(Bhi, BASE, 1); MMQ(Alo, Blo, 0, 0); ORD(RB);                                     \
        RD_A(Ahi, BASE, 1); MMQ(Alo, Bhi, 0, 1); ORD(RA);                                     \
        PINB(Bhi, NB); PIN(Ahi, NA);                                                          \
        BARRIER;                     /* every read of this stage is done; the next has landed */ \
        RD_A(Alo, NEXT, 0); MMQ(Ahi, Bhi, 1, 1); ORD(RA);                                     \
        RD_B(Bhi, NEXT, 1); MMQ(Ahi, Blo, 1, 0); ORD(RB);                                     \
    } while (0)
#define PL_STAGE_ODD(BASE, NEXT, BARRIER)                                                     \
    do {                                                                                      \
        RD_B(Blo, BASE, 0); MMQ(Alo, Bhi, 0, 1); ORD(RB);                                     \
        RD_A(Ahi, BASE, 1); MMQ(Alo, Blo, 0, 0); ORD(RA);                                     \
        PINB(Blo, NB); PIN(Ahi, NA);                                                          \
        BARRIER;                                                                              \
        RD_A(Alo, NEXT, 0); MMQ(Ahi, Blo, 1, 0); ORD(RA);                                     \
        RD_B(Blo, NEXT, 0); MMQ(Ahi, Bhi, 1, 1); ORD(RB);                                     \
    } while (0)
// __syncthreads() fences: it waits for EVERY outstanding memory operation of the wave (stores flushed at a seam of a
// balanced launch included).  The stage barrier only has to order LDS traffic: the fragment reads of the stage have
// returned (PIN), so wait for LDS and meet.
#define PL_RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifndef PL_MAIN_BARRIER
#ifdef MDBN_STAMP
#define PL_MAIN_BARRIER() do { const long long t_ = clock64(); __syncthreads(); bar_wait += clock64() - t_; } while (0)
#else
#define PL_MAIN_BARRIER() __syncthreads()
#endif
#endif
// NBH = 16-column blocks per B half of a wave's tile: 2 (64 x 64 per wave, 128-column tile) or 1 (64 x 32, 64-column tile)
#define PL_CONSUME_CONSTS()                                                                   \
    constexpr int NA = AP == 3 ? 3 : 1, NB = AP == 0 ? 1 : 3;      /* AP = 0: bf16-input reporting mode */ \
    constexpr int RA = 2 * NA * (LA == LAY_MN ? 2 : 1), RB = NBH * NB * (LB == LAY_MN ? 2 : 1), \
                  NM = 2 * NBH * (AP == 3 ? 6 : (AP == 1 ? 3 : 1));

template <int LA, int LB, int AP, int NBH = 2>
__device__ __forceinline__ void pl_consume16(char* smem, int nt, int lane, int wm, int wn, pf32x4a (&acc)[4][2 * NBH],
                                             bool open_barrier = true, long long* bar_wait_out = nullptr)
{
    long long bar_wait = 0;       // diagnostic builds: shader cycles spent at the stage barriers
    (void)bar_wait;
    int offA[4][2], offB[4][2];
    pl_offsets16<LA, LB>(lane, wm, wn, offA, offB);
    PL_CONSUME_CONSTS();
    pbf16x8 Alo[3][2], Ahi[3][2], Blo[3][2], Bhi[3][2];
    if (open_barrier) __syncthreads();               // stage 0 landed
    RD_A(Alo, smem, 0);
    RD_B(Blo, smem, 0);
    for (int it = 0; it < nt; it += 2) {
        {   // even stage
            const char* base = smem + (it % PL_NSTAGE) * PL_STAGE;
            const char* next = smem + ((it + 1) % PL_NSTAGE) * PL_STAGE;
            PL_STAGE_EVEN(base, next, PL_MAIN_BARRIER());
        }
        if (it + 1 < nt) {   // odd stage
            const char* base = smem + ((it + 1) % PL_NSTAGE) * PL_STAGE;
            const char* next = smem + ((it + 2) % PL_NSTAGE) * PL_STAGE;
            PL_STAGE_ODD(base, next, PL_MAIN_BARRIER());
        }
    }
    if (bar_wait_out) *bar_wait_out = bar_wait;
}

// FUSED: 0 = split-K slab / plain C store; 1 = activation + sampling epilogue on the parked tile (unsplit
// forward pass); 2 = statistics GEMM: finalize units on the ramp-up, parameter update (+ W planes) on the
// parked tile.  AP = planes of A (3, or 1 for 0/1 samples).
//
// (An in-launch split-K reduction of the forward passes was built three times and measured slower each time -- rounds 1-2:
// write-through slabs + last-arriver reduce / every workgroup reduces its rows, 222.9 / 178.6 against 152.0 us per step;
// round 3: the XCD-local form, a tile's split-K workgroups on one XCD, plain slab stores that stay in its L2, a per-launch
// placement proof, 173.9 against 154.7 us (profiles/r03a_xcd_local_inkernel_reduce_ab.log; the code is in the history at
// commit 23f393d).  The chain store -> drain -> arrive -> poll -> load -> activate costs what the kernel boundary costs,
// and the activation then runs on 512 threads per CU instead of 2048.  No kernel of this library waits for another
// workgroup.)
template <int LA, int LB, int AP, int FUSED, int MS, int BN = 128>
__global__ __launch_bounds__(64 * (4 + PL_LW)) void gemm_planes_kernel(PlaneGemmArgs g)
{
    static_assert(BN == 128 || (BN == 64 && MS == 16 && FUSED == 1 && LB == LAY_K), "64-column tiles: unsplit ROW-operand forward pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if PL_ARGS_EARLY
    // the kernel arguments every wave needs before its first global access, requested in ONE batch at the top: hipcc loads
    // kernarg fields where they are first used, and the head of this kernel was three scalar-load round trips in a row
    // (tile mapping, reduction extent, the operand pointers) before the first LDS-DMA could be issued
    asm volatile("" :: "s"(g.A), "s"(g.lda), "s"(g.pa), "s"(g.B), "s"(g.ldb), "s"(g.pb), "s"(g.C), "s"(g.ldc), "s"(g.slab_stride),
                 "s"(g.kchunk), "s"(g.tiles_m), "s"(g.tiles_n), "s"((int)gridDim.x));
#endif
    // XCD-aware (split, tile) order, as the other GEMM kernels: placement only affects speed
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qq = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.tiles_m <= g.tiles_n) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * 128, n0 = tn * BN;
    const int kbeg = ks * g.kchunk;
    const int nt = g.kchunk / 32;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    EarlySpeed es;
    (void)es;
    if (wave >= 4) {
        pl_loader<LA, LB, AP, MS, (FUSED == 2 || (FUSED == 0 && LA == LAY_MN && LB == LAY_MN && AP == 3)) && MS == 16, BN>(
            g, smem, wave - 4, lane, m0, n0, kbeg, nt, &es);
        if constexpr (FUSED == 0) return;
    } else {
        const int r = lane & 31, h = lane >> 5;
        const int wm = (wave >> 1) * 64, wn = (wave & 1) * (BN / 2);
        if constexpr (FUSED == 0 && LA == LAY_MN && LB == LAY_MN) {
            if (g.db.on) deferred_bias_update(g.db, wave, lane);     // data-parallel: bias / cost half of the previous step's update
        }
        if constexpr ((FUSED == 2 || FUSED == 0) && LA == LAY_MN && LB == LAY_MN) {
            if (g.fin_enabled) {        // statistics GEMM: finalize units while the first stages are in flight
                const int nu = fin_units(g.fin);
                for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += 4 * (int)gridDim.x)
                    finalize_unit(g.fin, unit, lane);
            }
        }
        if constexpr (MS == 16) {
            constexpr int NBH = BN / 64, NBB = 2 * NBH;          // 16-column blocks of a wave's tile
            pf32x4a acc[4][NBB];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NBB; ++b) acc[a][b] = pf32x4a{0.f, 0.f, 0.f, 0.f};
#ifdef MDBN_STAMP
#define PL_MSTAMP(SLOT)                                                                       \
    do {                                                                                      \
        if (PL_STAMP_COND && g.stamps && wave == 0 && lane == 0)                              \
            g.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64();                     \
    } while (0)
#else
#define PL_MSTAMP(SLOT) do {} while (0)
#endif
            PL_MSTAMP(8);
            __syncthreads();                         // (the barrier pl_consume16 opens with: stage 0 landed)
            PL_MSTAMP(9);
#ifdef MDBN_STAMP
            if (PL_STAMP_COND && g.stamps && wave == 0 && lane == 0)
                g.stamps[(int64_t)blockIdx.x * 16 + 13] = clock64();         // shader-clock cycles (s_memtime)
#endif
#ifdef MDBN_STAMP
            long long bw = 0;
            pl_consume16<LA, LB, AP, NBH>(smem, nt, lane, wm, wn, acc, false, &bw);
#else
            pl_consume16<LA, LB, AP, NBH>(smem, nt, lane, wm, wn, acc, false);
#endif
            PL_MSTAMP(10);
#ifdef MDBN_STAMP
            if (PL_STAMP_COND && g.stamps && wave == 0 && lane == 0) {
                g.stamps[(int64_t)blockIdx.x * 16 + 14] = clock64();
                g.stamps[(int64_t)blockIdx.x * 16 + 15] = bw;
            }
#endif
            // accumulator (16x16): col = lane & 15, row = 4 * (lane >> 4) + e
            const int c16 = lane & 15, q4 = lane >> 4;
            if constexpr (FUSED != 0) {
                float* T = reinterpret_cast<float*>(smem);
                constexpr int LDT = BN + 8;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < NBB; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            T[(wm + 16 * a + 4 * q4 + e) * LDT + wn + 16 * b + c16] = acc[a][b][e];
            } else {
                float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < NBB; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            C[(int64_t)(m0 + wm + 16 * a + 4 * q4 + e) * g.ldc + n0 + wn + 16 * b + c16] = acc[a][b][e];
                PL_MSTAMP(11);
#ifdef MDBN_STAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                PL_MSTAMP(12);
                return;
            }
        } else {
        // fragment byte offsets inside a plane image: [32-row block][k16 step][first / second tr read]
        int offA[2][2][2], offB[2][2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int gq = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
                {
                    const int row = wm + 32 * b + r;
                    if (LA == LAY_K) {
                        offA[b][s][0] = row * 64 + ((((2 * s + h) ^ ((row >> 2) & 3))) << 4);
                        offA[b][s][1] = 0;
                    } else {
                        const int ch = (wm + 32 * b) / 8 + 2 * gq + (p >> 1);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
                            offA[b][s][tt] = (16 * s + 8 * h + 4 * tt + q) * 256 + ((ch ^ (q << 2)) << 4) + 8 * (p & 1);
                    }
                }
                {
                    const int row = wn + 32 * b + r;
                    if (LB == LAY_K) {
                        offB[b][s][0] = row * 64 + ((((2 * s + h) ^ ((row >> 2) & 3))) << 4);
                        offB[b][s][1] = 0;
                    } else {
                        const int ch = (wn + 32 * b) / 8 + 2 * gq + (p >> 1);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
                            offB[b][s][tt] = (16 * s + 8 * h + 4 * tt + q) * 256 + ((ch ^ (q << 2)) << 4) + 8 * (p & 1);
                    }
                }
            }
        pf32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

        pbf16x8 f0a[3][2], f0b[3][2], f1a[3][2], f1b[3][2];
#define PL_FRAGS(FA, FB, BASE, S)                                                             \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                          \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                       \
            if (pl < AP) FA[pl][a] = pl_frag<LA>((BASE) + pl * PL_PLANE, offA[a][S][0], offA[a][S][1]); \
            FB[pl][a] = pl_frag<LB>((BASE) + (3 + pl) * PL_PLANE, offB[a][S][0], offB[a][S][1]); \
        }
#define PL_CHAIN(T, FA, FB, a, b)                                                                \
    do {                                                                                      \
        if constexpr (AP == 3) {                /* smallest products first */                 \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2][a], FB[0][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[0][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);      \
        } else {        /* A = a1 exactly: a1 b3 + a1 b2 + a1 b1 is the full product */       \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], T, 0, 0, 0);      \
            T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], T, 0, 0, 0);      \
        }                                                                                     \
    } while (0)
        // step-local accumulators on the propup layout, exactly as gemm_bf16x6_kernel (X6_STAGE_ACC): same bits
#define PL_MMA(FA, FB)                                                                        \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                             \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                       \
            if constexpr (PL_STAGE_ACC && LA == LAY_K && LB == LAY_MN) {                      \
                pf32x16 t_;                                                                   \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) t_[e] = 0.f;                   \
                PL_CHAIN(t_, FA, FB, a, b);                                                   \
                acc[a][b] += t_;                                                              \
            } else {                                                                          \
                PL_CHAIN(acc[a][b], FA, FB, a, b);                                            \
            }                                                                                 \
        }
        // issue order: the NEXT step's fragment reads interleaved one by one under this step's MFMAs
        constexpr int NRD = 2 * (AP * (LA == LAY_MN ? 2 : 1) + 3 * (LB == LAY_MN ? 2 : 1)), NMM = 4 * (AP == 3 ? 6 : 3);
#define PL_ORDER()                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < (NRD < NMM ? NRD : NMM); ++i_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
    }                                                                                         \
    if (NRD > NMM) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMM, 0);                 \
    if (NMM > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
        __syncthreads();                             // stage 0 landed
        PL_FRAGS(f0a, f0b, smem, 0);
        // The F0 reload at the end of the body is unconditional (on the last stage it reads a slot that is no
        // longer written and the values are never used): see the note on the guarded reload in DESIGN.md.
        for (int it = 0; it < nt; ++it) {
            const char* base = smem + (it % PL_NSTAGE) * PL_STAGE;
            const char* next = smem + ((it + 1) % PL_NSTAGE) * PL_STAGE;
            PL_FRAGS(f1a, f1b, base, 1);
            PL_MMA(f0a, f0b);
            PL_ORDER();
            // pin the F1 reads before the barrier (hipcc may sink LDS loads past s_barrier when their only use is in
            // a later basic block: the root cause of the "guarded reload" miscompile)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (pl < AP) asm volatile("" :: "v"(f1a[pl][a]));
                    asm volatile("" :: "v"(f1b[pl][a]));
                }
            __syncthreads();                         // every read of stage `it` is done; stage it + 1 has landed
            PL_FRAGS(f0a, f0b, next, 0);
            PL_MMA(f1a, f1b);
            PL_ORDER();
        }
#undef PL_FRAGS
#undef PL_MMA
#undef PL_CHAIN
#undef PL_ORDER
        // accumulator (32x32): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
        if constexpr (FUSED != 0) {
            float* T = reinterpret_cast<float*>(smem);
            constexpr int LDT = 128 + 8;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h) * LDT + wn + 32 * b + r] = acc[a][b][e];
        } else {
            float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = n0 + wn + 32 * b + r;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                        C[(int64_t)row * g.ldc + col] = acc[a][b][e];
                    }
                }
            return;
        }
            }
    }
    if constexpr (FUSED != 0) {     // all 8 waves work on the parked tile (every DMA has landed: the loaders drained vmcnt)
        __syncthreads();
        float* T = reinterpret_cast<float*>(smem);
        constexpr int NT = 64 * (4 + PL_LW);
#ifdef MDBN_STAMP
        if (PL_STAMP_COND && g.stamps && threadIdx.x == 0) g.stamps[(int64_t)blockIdx.x * 16 + 11] = wall_clock64();
#endif
#ifdef MDBN_STAMP
        if constexpr (FUSED == 1 && BN == 64)
            fused_tile_epilogue_4x4<NT>(g.epi, T, m0, n0, (PL_STAMP_COND && g.stamps) ? g.stamps + (int64_t)blockIdx.x * 16 + 5 : nullptr);
        else
#endif
        if constexpr (FUSED == 1 && BN == 64) fused_tile_epilogue_4x4<NT>(g.epi, T, m0, n0);     // one pass, loads up front
        else if constexpr (FUSED == 1) fused_tile_epilogue<128, BN, NT>(g.epi, T, m0, n0);
        else if (MS == 16 && g.upd.early == 1) {     // W went early (pl_loader); the loader waves hold speed_old (+ W0)
            if (wave >= 4) {
                const int lt = (wave - 4) * 64 + lane;
                constexpr int LDT = 128 + 8;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int row = 8 * j + (lt >> 5), c4 = lt & 31;
                    const float4 st = *reinterpret_cast<const float4*>(T + row * LDT + 4 * c4);
                    float4 sn;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        setc(sn, c, upd_speed(upd_grad(comp(st, c), g.upd.inv_bs, g.upd.wc, comp(es.w0[j], c)), comp(es.sp[j], c), g.upd.mu));
                    *reinterpret_cast<float4*>(g.upd.Ws + (int64_t)(m0 + row) * g.upd.ld + n0 + 4 * c4) = sn;
                }
            }
        }
        else fused_update_epilogue<128, 128, NT>(g.upd, T, m0, n0);
#ifdef MDBN_STAMP
        if (PL_STAMP_COND && g.stamps && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g.stamps[(int64_t)blockIdx.x * 16 + 12] = wall_clock64();
        }
#endif
    }
}

template <int LA, int LB, int AP, int FUSED, int MS, int BN = 128>
static hipError_t launch_planes_m(const PlaneGemmArgs& g, hipStream_t s)
{
    auto kern = gemm_planes_kernel<LA, LB, AP, FUSED, MS, BN>;
    static bool attr_set = false;
    constexpr int lds = PL_NSTAGE * PL_STAGE;        // 144 KB (the parked tile of the epilogues, 70 KB, reuses it)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.splitk), dim3(64 * (4 + PL_LW)), lds, s, g);
    return hipGetLastError();
}

template <int LA, int LB, int AP, int FUSED>
static hipError_t launch_planes_t(const PlaneGemmArgs& g, hipStream_t s)
{
    if constexpr (AP == 0) return launch_planes_m<LA, LB, 0, FUSED, 16>(g, s);        // reporting mode: 16x16x32 only
    else return g.ms == 32 ? launch_planes_m<LA, LB, AP, FUSED, 32>(g, s) : launch_planes_m<LA, LB, AP, FUSED, 16>(g, s);
}

hipError_t launch_gemm_planes(int la, int lb, const PlaneGemmArgs& g, hipStream_t s)
{
    if (g.bn == 64) {       // unsplit forward pass on 128 x 64 tiles with the fused activation epilogue (propdown)
        if (g.M % 128 || g.N % 64 || g.K % 32 || g.kchunk != g.K || g.splitk != 1 || g.tiles_m != g.M / 128 || g.tiles_n != g.N / 64 ||
            (g.lda & 7) || (g.ldb & 7) || g.fused != 1 || la != LAY_K || lb != LAY_K || g.ms != 16 || (g.ap != 1 && g.ap != 3))
            return hipErrorInvalidValue;
        return g.ap == 1 ? launch_planes_m<LAY_K, LAY_K, 1, 1, 16, 64>(g, s) : launch_planes_m<LAY_K, LAY_K, 3, 1, 16, 64>(g, s);
    }
    if (g.M % 128 || g.N % 128 || g.kchunk % 32 || g.kchunk * g.splitk != g.K || g.tiles_m != g.M / 128 ||
        g.tiles_n != g.N / 128 || (g.lda & 7) || (g.ldb & 7) || ((g.fused == 1 || g.fused == 2) && g.splitk != 1) ||
        (g.ap != 0 && g.ap != 1 && g.ap != 3))
        return hipErrorInvalidValue;
#define PL_CASE(LAV, LBV, APV, FV) \
    if (la == LAV && lb == LBV && g.ap == APV && g.fused == FV) return launch_planes_t<LAV, LBV, APV, FV>(g, s)
    // propup: x planes (ROW) x W planes (COL)
    PL_CASE(LAY_K, LAY_MN, 3, 0); PL_CASE(LAY_K, LAY_MN, 3, 1); PL_CASE(LAY_K, LAY_MN, 1, 0); PL_CASE(LAY_K, LAY_MN, 1, 1);
    // propdown: h planes (ROW) x W planes (ROW)
    PL_CASE(LAY_K, LAY_K, 3, 0); PL_CASE(LAY_K, LAY_K, 3, 1); PL_CASE(LAY_K, LAY_K, 1, 0); PL_CASE(LAY_K, LAY_K, 1, 1);
    // statistics: [v0; nv]^T planes (COL) x [ph; -nh] planes (COL)
    PL_CASE(LAY_MN, LAY_MN, 3, 0); PL_CASE(LAY_MN, LAY_MN, 3, 2);
    // bf16-input reporting mode (one product; never used for parity)
    PL_CASE(LAY_K, LAY_MN, 0, 0); PL_CASE(LAY_K, LAY_MN, 0, 1); PL_CASE(LAY_K, LAY_K, 0, 0); PL_CASE(LAY_K, LAY_K, 0, 1);
    PL_CASE(LAY_MN, LAY_MN, 0, 0); PL_CASE(LAY_MN, LAY_MN, 0, 2);
#undef PL_CASE
    return hipErrorInvalidValue;
}

// ==================================================================================
// BALANCED launches: the same GEMM on ANY number of workgroups (data-parallel mode).
//
// A collective's kernel that runs beside the step takes whole CUs (RCCL's gfx950 all-reduce kernel: 248-256 VGPRs per
// wave, 37.6 KB LDS, 512 threads -- it cannot share a CU with a 144-KB GEMM block), and a 256-workgroup grid on 255 CUs
// runs in two rounds: measured with a stand-in kernel of that footprint, the c2 data-parallel step goes from 168 to
// 220 us as soon as EIGHT CUs are taken (scripts/dp_contention_probe.py, profiles/r02i_*).  So in data-parallel mode
// the GEMMs are launched on P = CUs - comm_cus workgroups and the work is cut evenly:
//   * F = tiles / P WHOLE tiles per workgroup ([w F, (w + 1) F)), swept from stage 0 by every workgroup in step;
//   * the stages of the R = tiles - F P tiles left over, in (tile, stage) order, cut into equal contiguous shares
//     ("stream-K" on the remainder only; a share is shorter than one tile, so at most two pieces).  With fewer tiles
//     than workgroups (the forward passes at c2) everything is remainder.
// The LDS ring and the loader waves run straight through the seams between segments; the MFMA waves flush their
// accumulators at each.  A shared tile's pieces are combined OUTSIDE the launch, by a kernel that follows it:
//   FIX = 0 (forward passes): every piece goes to its own slab, slab index = position of the workgroup among those
//           that share the tile; the activation epilogue launch sums the slabs a tile HAS (bal_tile_slabs), in order.
//   FIX = 1 (statistics GEMM, result wanted in place): whole tiles are stored where they belong; pieces of shared
//           tiles are parked in scratch (64 KB per workgroup, in accumulator-register layout: 1 KiB per store
//           instruction) and bal_fixup_kernel adds the pieces of each shared tile in workgroup order.
// No workgroup ever waits for another: an earlier version let a tile's owner collect the parked pieces inside the
// launch behind per-wave flags; the owner's serial, latency-bound collection (4 round trips to memory per piece) cost
// more than the extra launch (statistics GEMM 63-76 us against 52 + 4), and a spin-wait is a liability beside a
// collective that takes CUs away.
// Determinism: the cut depends only on (tiles, stages, P); pieces are added in a fixed order.
// A piece that does not reach its tile's last stage is computed FIRST, the piece that does LAST: every workgroup then
// starts at stage 0 of some tile, and the workgroups sweep the reduction index nearly in step -- that keeps the
// operand stages shared in L2 (ranges starting at scattered stages cost up to 60 %: statistics GEMM 63 -> 102 us).
// ==================================================================================
struct BalRange {
    int F, Tf;                    // whole tiles per workgroup; first remainder tile
    int tA, sA, tB, eB, nrem;     // remainder units [tA * S + sA, tB * S + eB) in remainder-tile space, eB in 1..S; pieces
    int pre, nseg, nt;            // remainder pieces processed BEFORE the whole tiles; segments; stages in all
};
__host__ __device__ inline BalRange bal_range(int w, int P, int tiles, int S)
{
    BalRange r;
    r.F = tiles / P; r.Tf = r.F * P;
    const int64_t Ur = (int64_t)(tiles - r.Tf) * S;
    const int Pr = bal_rem_blocks(tiles, S, P);
    const int u0 = w < Pr ? bal_first_unit(w, Pr, Ur) : 0, u1 = w < Pr ? bal_first_unit(w + 1, Pr, Ur) : 0;
    r.tA = r.sA = r.tB = r.eB = r.nrem = r.pre = 0;
    if (u1 > u0) {
        r.tA = u0 / S; r.sA = u0 - r.tA * S;
        r.tB = (u1 - 1) / S; r.eB = u1 - r.tB * S;
        r.nrem = r.tB - r.tA + 1;
        r.pre = r.nrem == 2 ? 1 : (r.eB != S ? 1 : 0);
    }
    r.nseg = r.F + r.nrem;
    r.nt = r.F * S + (u1 - u0);
    return r;
}
// segment k of a workgroup in processing order: (tile, first stage, end stage)
__host__ __device__ inline void bal_segment(const BalRange& r, int w, int k, int S, int& tile, int& s0, int& s1)
{
    if (k < r.pre) {                              // the piece that does not reach its tile's end
        if (r.nrem == 2) { tile = r.Tf + r.tB; s0 = 0; s1 = r.eB; }
        else { tile = r.Tf + r.tA; s0 = r.sA; s1 = r.eB; }
    } else if (k < r.pre + r.F) {
        tile = w * r.F + (k - r.pre); s0 = 0; s1 = S;
    } else {                                      // the piece that ends its tile
        tile = r.Tf + r.tA; s0 = r.sA; s1 = S;
    }
}

template <int LA, int LB, int AP>
__device__ __forceinline__ void pl_loader_bal(const PlaneGemmArgs& g, char* smem, int w, int lane, int wg, const BalRange& r, int nt, int S)
{
    constexpr int NA = AP == 3 ? 3 : 1, NB = AP == 0 ? 1 : 3;
    constexpr int NQ = 8 * (NA + NB), PER = NQ / PL_LW, PERA = 8 * NA / PL_LW;     // j < PERA: an A instruction
    const char* base[PER];        // tile (0, 0), stage 0
    const char* src[PER];
    unsigned dst[PER];
    int64_t step[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = w + j * PL_LW;
        const bool isA = j < PERA;
        const int plane = isA ? q / 8 : (q - 8 * NA) / 8, sub = q & 7;
        const int lay = isA ? LA : LB;
        const unsigned short* b0 = isA ? g.A + plane * g.pa : g.B + plane * g.pb;
        const int64_t ld = isA ? g.lda : g.ldb;
        dst[j] = (isA ? plane : 3 + plane) * PL_PLANE + sub * 1024;
        if (lay == LAY_K) {
            const int row = 16 * sub + (lane >> 2);
            const int c = (lane & 3) ^ pl_row_swz<16>(row);
            base[j] = reinterpret_cast<const char*>(b0 + (int64_t)row * ld + 8 * c);
            step[j] = 64;
        } else {
            const int k = 4 * sub + (lane >> 4);
            const int ch = (lane & 15) ^ pl_col_swz<16>(k);
            base[j] = reinterpret_cast<const char*>(b0 + (int64_t)k * ld + 8 * ch);
            step[j] = 64 * ld;
        }
    }
    int seg = 0, left = 0;
    auto next_segment = [&]() {       // point src[] at the first stage of segment `seg`
        int t, s0, s1;
        bal_segment(r, wg, seg, S, t, s0, s1);
        ++seg; left = s1 - s0;
        int tm, tn;
        if (g.tiles_m <= g.tiles_n) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
        else { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
        const int64_t offA = LA == LAY_K ? (int64_t)tm * 128 * g.lda * 2 : (int64_t)tm * 256;
        const int64_t offB = LB == LAY_K ? (int64_t)tn * 128 * g.ldb * 2 : (int64_t)tn * 256;
#pragma unroll
        for (int j = 0; j < PER; ++j) src[j] = base[j] + (j < PERA ? offA : offB) + (int64_t)s0 * step[j];
    };
    next_segment();
#define PL_ISSUE(T)                                                                           \
    do {                                                                                      \
        const unsigned so = ((T) % PL_NSTAGE) * PL_STAGE;                                     \
        _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                     \
            pl_glds16(src[j], so + dst[j], smem);                                             \
            src[j] += step[j];                                                                \
        }                                                                                     \
        if (--left == 0 && seg < r.nseg) next_segment();                                      \
    } while (0)
    PL_ISSUE(0);                                     // stage 0 alone first (see pl_loader)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // stage 0 landed
    if (nt > 1) { PL_ISSUE(1); }
    if (nt > 2) { PL_ISSUE(2); }
    int it = 0;
#define LD_SYNC()                                                                             \
    do {                                                                                      \
        if (it + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");          \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
        __builtin_amdgcn_s_barrier();                /* every read of stage `it` is done: its slot is free */ \
    } while (0)
    if constexpr (LA == LAY_MN && LB == LAY_MN && AP == 3) {
        if (g.upd.early == 2) {
            // Data-parallel step on balanced launches: the WHOLE deferred update of the previous step (phase 3), as in
            // pl_loader's early == 2 branch -- same loads one stage ahead behind counted waits, same arithmetic -- but over a
            // FLAT share of the [rows * ld] arrays instead of the workgroup's tile (a balanced workgroup has no tile of its
            // own; nothing of the update depends on this GEMM): 16 items of two 16-byte pieces per thread, pieces
            // [wg * flat_per_wg, (wg + 1) * flat_per_wg) (host: flat_per_wg <= 16 * 512, every workgroup has >= 20 stages).
            // Loads are unconditional at clamped, valid addresses; only the stores are predicated.
#define ASM_LOAD4(DST, PTR) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(DST) : "v"(PTR) : "memory")
#define ASM_WAIT(N) do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
            const int lt = w * 64 + lane;
            const float decay = upd_decay(g.upd.lr, g.upd.l2);
            const bool has_wc = g.upd.wc != 0.0f;
            const float* w0base = has_wc ? g.upd.W0 : g.upd.Ws;
            const int64_t total4 = (int64_t)g.upd.rows * g.upd.ld / 4;
            const int64_t base4 = (int64_t)wg * g.upd.flat_per_wg + lt;
            const int64_t end4 = total4 < (int64_t)(wg + 1) * g.upd.flat_per_wg ? total4 : (int64_t)(wg + 1) * g.upd.flat_per_wg;
            pf32x4 dw[16][2], ds[16][2], dt[16][2], d0[16][2];
#define BAL_ITEM_LOAD(I)                                                                      \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                        \
        const int64_t f_ = base4 + (2 * (I) + h_) * 256;                                      \
        const int64_t o_ = 4 * (f_ < total4 ? f_ : total4 - 1);                               \
        ASM_LOAD4(dw[I][h_], g.upd.W + o_); ASM_LOAD4(ds[I][h_], g.upd.Ws + o_);              \
        ASM_LOAD4(dt[I][h_], g.upd.Sprev + o_); ASM_LOAD4(d0[I][h_], w0base + o_);            \
    }
            BAL_ITEM_LOAD(0)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                LD_SYNC();
                if (i + 1 < 16) { BAL_ITEM_LOAD(i + 1) }
                PL_ISSUE(it + 3);
                if (i + 1 < 16) ASM_WAIT(PER + 8); else ASM_WAIT(PER);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int64_t f = base4 + (2 * i + h) * 256;
                    float4 sn, wn;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float w0j = has_wc ? d0[i][h][j] : 0.0f;
                        const float snj = upd_speed(upd_grad(dt[i][h][j], g.upd.inv_bs, g.upd.wc, w0j), ds[i][h][j], g.upd.mu);
                        setc(sn, j, snj);
                        setc(wn, j, upd_param(dw[i][h][j], decay, snj, g.upd.lr));
                    }
                    if (f < end4) {
                        *reinterpret_cast<float4*>(g.upd.Ws + 4 * f) = sn;
                        *reinterpret_cast<float4*>(g.upd.W + 4 * f) = wn;
                        if (g.upd.Wp) store_planes4(g.upd.Wp, g.upd.wp_stride, 4 * f, wn);
                    }
                }
                ++it;
            }
#undef BAL_ITEM_LOAD
#undef ASM_LOAD4
#undef ASM_WAIT
        }
    }
    for (; it < nt; ++it) {
        LD_SYNC();
        if (it + 3 < nt) { PL_ISSUE(it + 3); }
    }
#undef LD_SYNC
#undef PL_ISSUE
}

typedef unsigned int pu32x4b __attribute__((ext_vector_type(4)));

template <int LA, int LB, int AP, int FIX>
__global__ __launch_bounds__(64 * (4 + PL_LW)) void gemm_planes_bal_kernel(PlaneGemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = gridDim.x;
    // Hardware ids are dealt round-robin over the 8 XCDs, each with its own L2.  xcd_group = 1: consecutive logical
    // workgroups (neighbouring tiles: shared operand tiles) sit on the same XCD (propdown at c2: 25.7 -> 21.0 us);
    // xcd_group = 0: consecutive workgroups (the shares of ONE tile, which have no operand stage in common) are dealt
    // over the XCDs, so that an XCD sees one slice of the reduction index of every tile (propup at c2: 30.6 -> 27.1 us)
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3, qq = P >> 3, rem = P & 7;
    const int w = g.xcd_group ? (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot : bid;
    const int S = g.K / 32, tiles = g.tiles_m * g.tiles_n;
    const BalRange r = bal_range(w, P, tiles, S);
    const int nt = r.nt;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (nt <= 0) return;                             // more workgroups than stages (the host never launches that)
    if (wave >= 4) {
        pl_loader_bal<LA, LB, AP>(g, smem, wave - 4, lane, w, r, nt, S);
        return;
    }
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    if constexpr (LA == LAY_MN && LB == LAY_MN) {
        if (g.db.on) deferred_bias_update(g.db, wave, lane);     // data-parallel: bias / cost half of the previous step's update
        if (g.fin_enabled) {            // statistics GEMM: finalize units while the first stages are in flight
            const int nu = fin_units(g.fin);
            for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += 4 * (int)gridDim.x)
                finalize_unit(g.fin, unit, lane);
        }
    }
    int offA[4][2], offB[4][2];
    pl_offsets16<LA, LB>(lane, wm, wn, offA, offB);
    constexpr int NBH = 2;
    PL_CONSUME_CONSTS();
    pbf16x8 Alo[3][2], Ahi[3][2], Blo[3][2], Bhi[3][2];
    pf32x4a acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = pf32x4a{0.f, 0.f, 0.f, 0.f};
    const int c16 = lane & 15, q4 = lane >> 4;
    // buffer descriptors: the matrix side (slabs, or the result in place) and the scratch of the in-place variant;
    // per-lane offsets are 32-bit, the (a, b, e) part of an offset is scalar
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)g.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(g.scratch, 0, FIX ? 2 * P * 65536 : 0, 0x00020000);
    const int ldc4 = (int)g.ldc * 4;
    const int lane_off = (wm + 4 * q4) * ldc4 + (wn + c16) * 4;

    int it = 0, seg = 0, left, t_cur, s_beg, s_end;
    bal_segment(r, w, seg, S, t_cur, s_beg, s_end);
    left = s_end - s_beg;
    // the segment [s_beg, s_end) of tile t_cur is complete: flush the accumulators
    auto flush = [&]() __attribute__((always_inline)) {
        int tm, tn;
        if (g.tiles_m <= g.tiles_n) { tn = t_cur / g.tiles_m; tm = t_cur - tn * g.tiles_m; }
        else { tm = t_cur / g.tiles_n; tn = t_cur - tm * g.tiles_n; }
        const bool shared = s_beg != 0 || s_end != S;
        int base_off = tm * 128 * ldc4 + tn * 512;                           // bytes; < 2^31 (host check)
        if (FIX == 0 && t_cur >= r.Tf) {
            const int Pr = bal_rem_blocks(tiles, S, P);
            const int64_t Ur = (int64_t)(tiles - r.Tf) * S;
            base_off += (w - bal_block_of((int64_t)(t_cur - r.Tf) * S, Pr, Ur)) * (int)(g.slab_stride * 4);
        }
        if (FIX != 0 && shared) {
            // a piece of a shared tile: parked in register layout, slot 2 w + (0: the piece that ends its tile, 1: the
            // other one), [wave][a][b][lane] x 16 bytes
            const int slot = 2 * w + (s_end != S ? 1 : 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4b, acc[a][b]), srsrc, lane * 16,
                                                           ((slot * 4 + wave) * 16 + a * 4 + b) * 1024, 0);
        } else {
            // accumulator (16x16): col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = acc[a][b][e];      // (a bit_cast of the vector-element lvalue itself reads element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), crsrc, lane_off,
                                                              base_off + (16 * a + e) * ldc4 + 64 * b, 0);
                    }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = pf32x4a{0.f, 0.f, 0.f, 0.f};
        if (++seg < r.nseg) {
            bal_segment(r, w, seg, S, t_cur, s_beg, s_end);
            left = s_end - s_beg;
        }
    };

    __syncthreads();                                 // stage 0 landed
    RD_A(Alo, smem, 0);
    RD_B(Blo, smem, 0);
    while (it < nt) {
        {
            const char* base = smem + (it % PL_NSTAGE) * PL_STAGE;
            const char* next = smem + ((it + 1) % PL_NSTAGE) * PL_STAGE;
            PL_STAGE_EVEN(base, next, PL_RAW_BARRIER());
            ++it;
            if (--left == 0) flush();
        }
        if (it < nt) {
            const char* base = smem + (it % PL_NSTAGE) * PL_STAGE;
            const char* next = smem + ((it + 1) % PL_NSTAGE) * PL_STAGE;
            PL_STAGE_ODD(base, next, PL_RAW_BARRIER());
            ++it;
            if (--left == 0) flush();
        }
    }
}

// Shared tiles of an in-place balanced launch: C tile = sum of the parked pieces in workgroup order.  One workgroup
// per (shared tile, 16-row band a of each wave's quadrant): thread = (wave, lane) of the GEMM's MFMA waves, so the
// scratch is read in the layout it was written in (16 bytes per lane, 1 KiB per wave-instruction).
__global__ __launch_bounds__(256) void bal_fixup_kernel(const float* __restrict__ scratch, float* __restrict__ C, int64_t ldc,
                                                       int tiles_m, int tiles_n, int S, int P)
{
    const int tiles = tiles_m * tiles_n, Tf = (tiles / P) * P;
    const int t = Tf + (int)blockIdx.x / 4, a = (int)blockIdx.x & 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Pr = bal_rem_blocks(tiles, S, P);
    const int64_t Ur = (int64_t)(tiles - Tf) * S, u0 = (int64_t)(t - Tf) * S;
    const int wfirst = bal_block_of(u0, Pr, Ur), wlast = bal_block_of(u0 + S - 1, Pr, Ur);
    if (wfirst == wlast) return;                     // one workgroup computed the whole tile and stored it itself
    pf32x4a acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = pf32x4a{0.f, 0.f, 0.f, 0.f};
    for (int w = wfirst; w <= wlast; ++w) {
        // the piece of workgroup w on this tile: the one that ends the tile only for the last sharer
        const int slot = 2 * w + (w == wlast ? 0 : 1);
        const pf32x4a* p = reinterpret_cast<const pf32x4a*>(scratch) + (((int64_t)slot * 4 + wave) * 16 + a * 4) * 64 + lane;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] += p[b * 64];
    }
    int tm, tn;
    if (tiles_m <= tiles_n) { tn = t / tiles_m; tm = t - tn * tiles_m; }
    else { tm = t / tiles_n; tn = t - tm * tiles_n; }
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, c16 = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            C[(int64_t)(tm * 128 + wm + 16 * a + 4 * q4 + e) * ldc + tn * 128 + wn + 16 * b + c16] = acc[b][e];
}

template <int LA, int LB, int AP, int FIX>
static hipError_t launch_planes_bal_t(const PlaneGemmArgs& g, hipStream_t s)
{
    auto kern = gemm_planes_bal_kernel<LA, LB, AP, FIX>;
    static bool attr_set = false;
    constexpr int lds = PL_NSTAGE * PL_STAGE;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.bal), dim3(64 * (4 + PL_LW)), lds, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !FIX) return e;
    const int tiles = g.tiles_m * g.tiles_n, rem = tiles - (tiles / g.bal) * g.bal;
    if (rem > 0)
        hipLaunchKernelGGL(bal_fixup_kernel, dim3(4 * rem), dim3(256), 0, s, g.scratch, g.C, g.ldc, g.tiles_m, g.tiles_n, g.K / 32, g.bal);
    return hipGetLastError();
}

// host view of the cut (mdbn_bal_segment: tests check coverage, balance and slab bookkeeping without a GPU)
int bal_segment_host(int tiles, int S, int P, int w, int k, int* tile, int* s0, int* s1, int* nseg, int* nt)
{
    if (tiles < 1 || S < 1 || P < 1 || w < 0 || w >= P) return -1;
    const BalRange r = bal_range(w, P, tiles, S);
    *nseg = r.nseg; *nt = r.nt;
    if (k < 0 || k >= r.nseg) { *tile = *s0 = *s1 = -1; return 0; }
    bal_segment(r, w, k, S, *tile, *s0, *s1);
    return 0;
}

int bal_max_segments(int tiles, int S, int P)
{
    int mx = 1;
    for (int t = (tiles / P) * P; t < tiles; ++t) mx = std::max(mx, bal_tile_slabs(t, tiles, S, P));
    return mx;
}

// g.bal = number of workgroups; g.fused = 0 (slabs, one per piece of a tile) or 4 (in place: shared tiles through
// g.scratch, 128 KB per workgroup, and bal_fixup_kernel)
hipError_t launch_gemm_planes_bal(int la, int lb, const PlaneGemmArgs& g, hipStream_t s)
{
    const int64_t U = (int64_t)g.tiles_m * g.tiles_n * (g.K / 32);
    if (g.M % 128 || g.N % 128 || g.K % 32 || g.tiles_m != g.M / 128 || g.tiles_n != g.N / 128 || (g.lda & 7) || (g.ldb & 7) ||
        g.bal < 1 || g.bal > U / 4 || (g.fused != 0 && g.fused != 4) || (g.ap != 0 && g.ap != 1 && g.ap != 3))
        return hipErrorInvalidValue;
    if (g.fused == 4 && (g.scratch == nullptr || (int64_t)g.bal * 2 * 65536 >= (int64_t)1 << 31)) return hipErrorInvalidValue;
    if (g.c_bytes <= 0 || g.c_bytes >= (int64_t)1 << 31) return hipErrorInvalidValue;
#define PL_CASE(LAV, LBV, APV)                                                                \
    if (la == LAV && lb == LBV && g.ap == APV)                                                \
        return g.fused == 4 ? launch_planes_bal_t<LAV, LBV, APV, 1>(g, s) : launch_planes_bal_t<LAV, LBV, APV, 0>(g, s)
    PL_CASE(LAY_K, LAY_MN, 3); PL_CASE(LAY_K, LAY_MN, 1); PL_CASE(LAY_K, LAY_K, 3); PL_CASE(LAY_K, LAY_K, 1);
    PL_CASE(LAY_MN, LAY_MN, 3);
    PL_CASE(LAY_K, LAY_MN, 0); PL_CASE(LAY_K, LAY_K, 0); PL_CASE(LAY_MN, LAY_MN, 0);
#undef PL_CASE
    return hipErrorInvalidValue;
}

#undef RD_A
#undef RD_B
#undef MMQ
#undef MMQ_CHAIN
#undef ORD
#undef PIN
#undef PL_STAGE_EVEN
#undef PL_STAGE_ODD
#undef PL_RAW_BARRIER
#undef PL_CONSUME_CONSTS

// ----------------------------------------------------------------------------------
// f32 matrix -> three bf16 planes (same shape, same ld): W once per externally written W, and any tensor a
// caller hands in as f32.  One float4 per thread.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* __restrict__ X, int64_t n4,
                                                           unsigned short* __restrict__ P, int64_t plane_stride)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        store_planes4(P, plane_stride, i * 4, X[i]);
}

hipError_t launch_split_planes(const float* X, int64_t rows, int64_t ld, unsigned short* P, int64_t plane_stride, hipStream_t s)
{
    const int64_t n4 = rows * ld / 4;
    if (n4 <= 0) return hipSuccess;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(X), n4, P, plane_stride);
    return hipGetLastError();
}

// minibatch gather (dbn.py:307) that also writes the rows' planes: dst f32 [n_idx][ld] (nullable) + planes [3][.][ld].
// A thread moves 8 columns (two float4 in, two float4 + three 16-byte plane stores out) of GR consecutive minibatch
// rows, all its loads issued before the first store.  GR = 1: more rows per thread change nothing on the step
// (same-box builds, scripts/build_variants.py: 149.2 / 150.6 / 150.2 / 150.0 us at GR = 1 / 2 / 4 / 8).
#ifndef MDBN_GATHER_ROWS
#define MDBN_GATHER_ROWS 1
#endif
constexpr int GR = MDBN_GATHER_ROWS;
__global__ __launch_bounds__(256) void gather_planes_kernel(const float* __restrict__ src, int64_t n_rows, int64_t ld_src,
                                                            const void* __restrict__ idx, int idx64, int64_t ld8, int64_t n_idx,
                                                            float* __restrict__ dst, int64_t ld_dst,
                                                            unsigned short* __restrict__ P, int64_t plane_stride)
{
    const int64_t r0 = (int64_t)blockIdx.y * GR;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld8) return;
    float4 v[GR][2];
#pragma unroll
    for (int i = 0; i < GR; ++i) {
        const int64_t r = r0 + i < n_idx ? r0 + i : n_idx - 1;          // clamped: the value is not stored
        int64_t s = r;
        if (idx) s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
        if (s < 0) s += n_rows;
        s = s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);
        v[i][0] = reinterpret_cast<const float4*>(src + s * ld_src)[2 * c];
        v[i][1] = reinterpret_cast<const float4*>(src + s * ld_src)[2 * c + 1];
    }
#pragma unroll
    for (int i = 0; i < GR; ++i) {
        const int64_t r = r0 + i;
        if (r >= n_idx) break;
        if (dst) {                                    // the float32 copy is optional (mdbn_cd_args.keep_f32)
            reinterpret_cast<float4*>(dst + r * ld_dst)[2 * c] = v[i][0];
            reinterpret_cast<float4*>(dst + r * ld_dst)[2 * c + 1] = v[i][1];
        }
        const float x[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
        unsigned short q[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(x[j], q[0][j], q[1][j], q[2][j]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint4 w;
            w.x = q[p][0] | ((unsigned)q[p][1] << 16); w.y = q[p][2] | ((unsigned)q[p][3] << 16);
            w.z = q[p][4] | ((unsigned)q[p][5] << 16); w.w = q[p][6] | ((unsigned)q[p][7] << 16);
            *reinterpret_cast<uint4*>(P + p * plane_stride + r * ld_dst + 8 * c) = w;
        }
    }
}

hipError_t launch_gather_planes(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src, const void* idx, int idx64,
                                int64_t n_idx, float* dst, int64_t ld_dst, unsigned short* P, int64_t plane_stride, hipStream_t s)
{
    if (n_idx <= 0) return hipSuccess;
    if (n_idx > 65535 || (cols_ld & 7)) return hipErrorInvalidValue;         // the plane path has ld % 128 == 0
    const int64_t ld8 = cols_ld >> 3;
    const int threads = ld8 >= 256 ? 256 : 64;
    hipLaunchKernelGGL(gather_planes_kernel, dim3((unsigned)((ld8 + threads - 1) / threads), (unsigned)((n_idx + GR - 1) / GR)),
                       dim3(threads), 0, s, src, n_rows, ld_src, idx, idx64, ld8, n_idx, dst, ld_dst, P, plane_stride);
    return hipGetLastError();
}

}  // namespace mdbn
