// Thin-batch CD-k step (mdbn_thin.hip): minibatches of <= 32 rows -- the reference trains every preset at batch_size = 20
// (MDBN.py:46, AMLsm2.py:245; RBM.training defaults to 10, rbm.py:484-491) -- on layers whose W does not fit one CU's LDS
// (784 -> 500, the 19 937-gene layer of AMLsm2.py:242-251).  At this batch the step is a stream over W: arguments,
// geometry and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdbn_kernels.h"

namespace mdbn {

constexpr int TH_NT = 512;                  // threads per workgroup of the pass kernels (8 waves)
constexpr int TH_UNT = 256;                 // ... of the update kernel (4 waves: two workgroups per CU overlap their phases)
constexpr int TH_MAXB = 32;                 // minibatch rows: one M tile of v_mfma_f32_32x32x16_bf16
constexpr int TH_MAX_LDS = 160 * 1024;
constexpr int TH_ACT_NT = 1024;             // threads of the partial-sum + activation kernel (16 waves share the partials)
constexpr int TH_MAX_RPW = 256;             // rows of W per workgroup of a pass: <= 8 tiles of 32 rows (one per wave in phase 1)

// Row pitch (floats) of the K-contiguous float32 LDS image of W, read with ds_read_b128 per lane and row: a multiple of 4
// with pitch / 4 odd, so that the 16 lanes of a read group touch 16 different bank quads.
__host__ __device__ inline int thin_pitch(int cols4x) { return ((cols4x >> 2) & 1) ? cols4x : cols4x + 4; }

struct ThinGeom {
    int Bq;                                 // minibatch rows rounded up to whole Philox blocks (4 rows)
    int G, rpw;                             // workgroups of a pass (= partials of [B, H]); rows of W per workgroup (contiguous ranges)
    int PW;                                 // pitch of the float32 image of W in LDS
    int nt2;                                // 32-column output tiles of the upward product per wave (1 | 2)
    int lds_pass, lds_up;                   // dynamic LDS of the down + up pass / of the up-only pass (bytes)
    int Gu, rpu;                            // workgroups / rows per workgroup of the update kernel
    int lds_upd;
    int lds_ahead;                          // dynamic LDS of the update + next positive phase kernel (0: it does not fit)
};

// bytes of LDS the down + up pass needs for rpw rows per workgroup (layout: thin_pass_kernel)
__host__ __device__ inline int thin_pass_lds(int rpw, int Bq, int ldh)
{
    const int R16 = (rpw + 15) & ~15, R32 = (rpw + 31) & ~31, K16 = (ldh + 15) & ~15;
    const int PW = thin_pitch(K16), PH = K16 + 8, PXb = R16 + 8;
    const int ntile1 = R32 >> 5, S1 = 8 / ntile1 > 0 ? 8 / ntile1 : 1;
    const int hk = Bq * PH * 2, red = S1 * Bq * R32 * 4;
    return R16 * PW * 4 + (((hk > red ? hk : red) + 15) & ~15) + 3 * Bq * PXb * 2;
}

// Does the thin path serve this shape?  B <= 32, ldh <= 512 (two 32-column tiles of the upward product per wave), and
// a block of >= 16 rows of W fits one CU's LDS beside the chain state.
__host__ __device__ inline bool thin_geom(int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int num_cu, ThinGeom& t)
{
    if (B < 1 || B > TH_MAXB || ldh > 512 || ldh % 4 || ldv % 4 || V < 1 || H < 1 || V > (int64_t)1 << 24) return false;
    t.Bq = (int)((B + 3) & ~int64_t(3));
    const int cus = num_cu > 0 ? num_cu : 1;
    int64_t rpw = (V + cus - 1) / cus;
    if (rpw < 16) rpw = 16;                 // (a narrow layer: fewer, longer workgroups = fewer partials to sum)
    if (rpw > TH_MAX_RPW) rpw = TH_MAX_RPW;
    while (rpw > 16 && thin_pass_lds((int)rpw, t.Bq, (int)ldh) > TH_MAX_LDS) rpw = ((rpw - 1) & ~int64_t(15)) > 16 ? ((rpw - 1) & ~int64_t(15)) : 16;
    if (thin_pass_lds((int)rpw, t.Bq, (int)ldh) > TH_MAX_LDS) return false;
    t.rpw = (int)rpw;
    t.G = (int)((V + rpw - 1) / rpw);
    t.PW = thin_pitch((int)((ldh + 15) & ~int64_t(15)));
    t.nt2 = (ldh + 31) / 32 <= 8 ? 1 : 2;
    t.lds_pass = thin_pass_lds(t.rpw, t.Bq, (int)ldh);
    t.lds_up = 3 * t.Bq * (((t.rpw + 15) & ~15) + 8) * 2;
#ifndef TH_UPD_WGS
#define TH_UPD_WGS 2        // update workgroups (256 threads) per CU: their phases overlap (A/B: profiles/r05i_thin_update_variants.log)
#endif
    int64_t rpu = (V + TH_UPD_WGS * cus - 1) / (TH_UPD_WGS * cus);
    if (rpu < 8) rpu = 8;
    while (rpu * 2 * t.Bq * 4 > 48 * 1024) rpu = (rpu + 1) / 2;
    t.rpu = (int)rpu;
    t.Gu = (int)((V + rpu - 1) / rpu);
    t.lds_upd = t.rpu * 2 * t.Bq * 4;
    {       // update(t) + positive phase(t + 1) in one pass over W: float32 image of the block | [v0; nv] block | planes of x'
        const int R16 = (t.rpw + 15) & ~15;
        const int need = R16 * t.PW * 4 + t.rpw * 2 * t.Bq * 4 + 3 * t.Bq * (R16 + 8) * 2;
        t.lds_ahead = need <= TH_MAX_LDS ? need : 0;
    }
    return true;
}

// One pass over W.  MODE 0 (positive phase, rbm.py:303): x = train_set_x[indexes] gathered by the kernel itself, partial
// sums of x W per workgroup.  MODE 1 (one gibbs_hvh, rbm.py:242-248 / :662-671): for its rows of W the workgroup computes
// v1 = act(h W^T + vbias) COMPLETE (it holds whole rows of W), applies the visible activation, and at once the partial
// sums of v1 W from the same rows in LDS: W is read from HBM once for both products.
struct ThinPassArgs {
    unsigned long long* stamps;             // diagnostic builds only (-DMDBN_STAMP): 16 wall-clock stamps per workgroup; else NULL
    int B, Bq, V, H;
    int64_t ldv, ldh;
    int G, rpw, PW;
    const float* W;
    float* part;                            // [G][Bq][ldh] partial sums of the upward product
    // MODE 0
    const float* data; int64_t n_data, ld_data;
    const void* idx; int idx64;
    float* v0_out;                          // [B][ldv] (V2 rows 0..B-1)
    // MODE 1
    const float* chain;                     // [B][ldh] hidden chain state
    const float* vbias;
    float* nv;                              // [B][ldv] visible mean (V2 rows B..2B-1)
    float* vs;                              // [B][ldv] Bernoulli visible sample (RBM) or NULL (GRBM)
    int gauss, last;
    const float* target; int64_t ld_target; // last step: reconstruction-cost target (v0)
    float* cost_partials;                   // last step: one per workgroup
    PhiloxKey rng;                          // .draw = 2t - 1
};

// sum of the G partials (float64, fixed order) + bias + activation + sampling: EpiArgs as act_epilogue_kernel
struct ThinActArgs {
    const float* part; int G, Bq;
    EpiArgs e;
};

// statistics + update for the workgroup's rows of W: S rows from [v0; nv]^T [ph; -nh] (rank 2B, rbm.py:411-412) formed in
// registers and consumed at once by the update rule (rbm.py:347-365): W and W_speed read and written once, S never stored
// (do_upd), or S / s_h / s_v / cost stored for a data-parallel all-reduce (!do_upd)
struct ThinUpdArgs {
    int B, Bq, V, H;
    int64_t ldv, ldh;
    int G, rpw;                             // workgroups / rows per workgroup of THIS kernel (ThinGeom.Gu, .rpu)
    const float* V2; const float* P2;       // [2B][ldv] = [v0; nv], [2B][ldh] = [ph; -nh]
    float* S; float* s_h; float* s_v; float* cost;      // packed statistics (S only when !do_upd)
    const float* cost_partials; int n_cost;
    int do_upd;
    UpdEpi upd;                             // W, Ws, W0, lr, l1, l2, wc, mu, inv_bs (Wp: planes of the new W or NULL)
    BiasUpd bu;
    // thin_update_ahead_kernel only (pass geometry in G / rpw): the positive phase of the NEXT minibatch from the rows of
    // W this pass has just updated -- x' = data[next_idx], its partials of x' W' (one per workgroup) and v0' into V2 rows 0..B-1
    const float* data; int64_t n_data, ld_data;
    const void* next_idx; int idx64;
    float* part_next;                       // [G][Bq][ldh]
    float* v0_next;                         // = V2 (rows 0..B-1 are overwritten AFTER this workgroup has read its block of them)
    int PW;
};

hipError_t launch_thin_pass(int mode, const ThinPassArgs& a, const ThinGeom& t, hipStream_t s);
hipError_t launch_thin_act(const ThinActArgs& a, hipStream_t s);
hipError_t launch_thin_update(const ThinUpdArgs& a, const ThinGeom& t, hipStream_t s);
hipError_t launch_thin_update_ahead(const ThinUpdArgs& a, const ThinGeom& t, hipStream_t s);

}  // namespace mdbn
