// Internal launch interface between the C-ABI (mdbn_capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "philox.h"
#include "../../include/mdbn_hip.h"

namespace mdbn {

enum { LAY_K = 0, LAY_MN = 1 };   // operand's fastest dimension: reduction index / output index

struct EpiArgs {
    const float* slabs;
    int64_t slab_stride;
    int nsplit;
    int rows, cols;        // logical extent of the output
    int64_t ld;            // leading dim of slabs and of pre/mean/sample
    const float* bias;
    float* pre;            // nullable
    float* mean;           // nullable; stored * mean_scale
    float* sample;         // nullable
    float mean_scale;
    int gauss;             // 0: sigmoid + Bernoulli ; 1: linear + N(0,1)
    const float* target;   // nullable: reconstruction-cost target (v0)
    int64_t ld_target;
    // target rows through a minibatch index (plane path without the float32 copy of v0: the target is the dataset
    // itself): row r of the output reads row target_idx[r] (numpy-style negative values, clamped like the gather)
    const void* target_idx;     // nullable: int32 / int64 [rows]
    int target_idx64;
    int64_t target_rows;        // rows of the matrix `target` points into
    float* cost_partials;  // nullable: one float per block
    float* colsum;         // nullable: [ceil(rows/4)][ld] partial column sums over each 4-row group
    int colsum_kind;       // 0: sum of stored (scaled) mean ; 1: sum of (target - mean) ; 2: sum of (target - sample)
    PhiloxKey rng;
    // bf16 plane outputs (mdbn_planes.hip): the stored (scaled) mean split exactly into three bf16 planes
    // [3][rows][ld] (plane p at mean_planes + p * plane_stride), the 0/1 sample as one bf16 plane
    unsigned short* mean_planes;   // nullable
    int64_t plane_stride;          // elements between planes
    unsigned short* sample_plane;  // nullable (Bernoulli samples only)
    // slabs written by a balanced GEMM (launch_gemm_planes_bal): a 128x128 tile has as many slabs as workgroups shared
    // its stages; bal_P = 0: every element has `nsplit` slabs
    int bal_P, bal_S, bal_tiles_m, bal_tiles_n;
};

// parameter update applied by the statistics GEMM to its own output tile (fused == 2)
struct UpdEpi {
    float* W; float* Ws; const float* W0;      // [rows][ld]; W0 nullable (then the live W)
    int64_t ld;
    int rows;
    float lr, l1, l2, wc, mu, inv_bs;
    unsigned short* Wp;            // nullable: bf16 planes [3][rows][ld] of the NEW W, kept in step with it
    int64_t wp_stride;
    const float* Sprev;            // early == 2: S block of the (all-reduced) statistics of the PREVIOUS step
    int64_t flat_per_wg;           // early == 2 in a BALANCED launch: 16-byte pieces of the flat [rows * ld] arrays per workgroup
    int early;                     // plane statistics GEMM: the PARAMETER half (W, W planes: it needs only the old W and the old
                                   // speed, rbm.py:364-365) is applied by the loader waves DURING the main loop; the epilogue
                                   // then only forms the new speed from the finished tile (needs l1 == 0 and wc == 0 or W0).
                                   // 2 (data-parallel, FUSED == 0): the loader waves apply the WHOLE deferred update of the
                                   // previous step (phase 3: speed' from Sprev, W' from speed') -- nothing of it depends on
                                   // this GEMM; the launch of update_kernel<true, true, true> disappears
};

// bias / cost half of that deferred update (update_kernel's leading blocks), run by the MFMA waves ahead of the main loop
struct DeferredBias {
    int on;
    float* hb; float* hbs; float* vb; float* vbs;
    const float* s_h; const float* s_v; const float* cost_sum;
    int64_t H, V;
    float lr, mu, inv_rows, cost_scale;
    float* cost_out;
};

// bias half of the update + monitoring cost, applied by finalize_stats_kernel
struct BiasUpd {
    float* hb; float* hbs; float* vb; float* vbs;
    int64_t H, V;
    float lr, mu, inv_rows, cost_scale;
    float* cost_out;
};

// bias statistics + cost total (+ bias half of the update): finalize_stats_kernel, or the consumer
// waves of the fused statistics GEMM during their ramp-up (fin_enabled)
struct FinArgs {
    const float* posP; const float* negP; const float* partV;   // [ngroups][ldh], [ngroups][ldh], [ngroups][ldv]
    int ngroups;
    int64_t ldh, ldv;
    const float* cost_partials; int n_cost;
    float* s_h; float* s_v; float* cost;
    BiasUpd bu; int do_bias;
};

struct GemmArgs {
    const float* A;        // LAY_K: [M][lda] ; LAY_MN: [K][lda]
    const float* B;        // LAY_K: [N][ldb] ; LAY_MN: [K][ldb]
    float* C;              // slabs: [splitk][M][ldc]
    int64_t lda, ldb, ldc;
    int64_t slab_stride;   // floats between consecutive split-K slabs
    int M, N, K;           // logical extents (loads beyond them read as zero)
    int Nst;               // columns stored (>= N; pad columns receive exact zeros)
    int kchunk;            // reduction extent per split, multiple of 32
    int splitk, tiles_m, tiles_n;
    int bn;                // block tile is 128 x bn (128 or 64)
    int bk;                // slice depth along the reduction index (32 or 64; 64 only with bn = 128)
    int inner_m;           // work-list order inside one split: 1 = tile_m fastest
    int x6;                // 1: bf16 matrix pipe at f32 accuracy (gemm_bf16x6_kernel; full 128x128 tiles only);
                           // 2: the same with a one-piece A operand (A holds 0/1 samples: 3 products instead of 6)
    int x6_pw;             // bf16x6: producer waves per operand (2 | 4)
    int cw;                // MFMA waves per SIMD (1 | 2; 2 only for unfused 128-column tiles)
    unsigned long long* stamps;   // diagnostic builds only (-DMDBN_STAMP); NULL otherwise
    int skinny;            // 1: skinny_gemm_kernel (tiles_n = 32-column strips, tiles_m = (32*mi)-row tiles,
                           //    splitk = K ranges, kchunk % 8 == 0)
    int mi;                // skinny: 32-row blocks per tile (1 | 2)
    int ni;                // streaming bf16x6 kernel (skinny = 1, x6 != 0): 32-column strips per tile (1 | 2)
    int fused;             // no split-K and an epilogue runs on the block's own output tile:
                           //   1 = activation (epi), 2 = parameter update (upd; statistics GEMM, C is not written)
    EpiArgs epi;           // fused == 1 (slabs / nsplit unused; one cost partial per block)
    UpdEpi upd;            // fused == 2
    int fin_enabled;       // fused == 2: the consumer waves also run the finalize units before the first slice
    FinArgs fin;
};



// GEMM on pre-split bf16 planes (mdbn_planes.hip): whole 128x128 tiles and 32-deep stages only
// Gather-ahead (statistics GEMM with the early parameter half): the loader waves also gather the NEXT minibatch's rows
// into the planes of the other X2 buffer, so the next step starts without a gather launch.
struct GatherAhead {
    const float* src; int64_t n_rows, ld_src;      // the training matrix
    const void* idx; int idx64;                    // [B] indices of the next minibatch (device); NULL = off
    int B, rpw, passes;                            // rows; rows per workgroup (<= 4); 256-octet passes per row
    unsigned short* P; int64_t plane_stride, ld;   // destination planes [3][.][ld] (rows 0..B-1 of the other buffer)
};

struct PlaneGemmArgs {
    const unsigned short* A; int64_t lda, pa;   // planes [ap][.][lda]; pa = elements between planes
    const unsigned short* B; int64_t ldb, pb;   // planes [3][.][ldb]
    float* C; int64_t ldc, slab_stride;         // fused == 0: slabs [splitk][M][ldc] (or plain C)
    int M, N, K, kchunk, splitk, tiles_m, tiles_n;
    int ap;                // planes of A: 3, or 1 when A holds 0/1 samples (three products instead of six);
                           // 0 = bf16-input reporting mode: the leading plane of each operand, ONE product
    int ms;                // MFMA shape: 16 = v_mfma_f32_16x16x32_bf16 (default), 32 = v_mfma_f32_32x32x16_bf16
    int bn;                // 0 / 128: 128 x 128 tiles; 64: 128 x 64 tiles (unsplit ROW-operand forward pass, fused == 1)
    int fused;             // 0 | 1 activation epilogue (epi) | 2 parameter update (upd) + finalize units (fin)
    // balanced launches (launch_gemm_planes_bal): `bal` workgroups share tiles x (K / 32) stages evenly;
    // fused = 0: slabs, one per piece of a tile; fused = 4: result in place, pieces of shared tiles through `scratch`
    unsigned long long* stamps;   // diagnostic builds only (-DMDBN_STAMP): 8 wall-clock stamps per workgroup (propup)
    int bal;
    int xcd_group;         // balanced: 1 = consecutive workgroups on one XCD, 0 = dealt over the XCDs
    float* scratch;        // fused == 4: 128 KB per workgroup
    int64_t c_bytes;       // balanced: bytes addressable from C (all slabs; < 2 GiB)
    EpiArgs epi;
    UpdEpi upd;
    int fin_enabled;
    FinArgs fin;
    GatherAhead ga;
    DeferredBias db;
};

// balanced launches (mdbn_planes.hip): U units in (tile, stage) order, workgroup w of P takes [w U / P, (w + 1) U / P)
__host__ __device__ inline int bal_first_unit(int w, int P, int64_t U) { return (int)(((int64_t)w * U) / P); }
__host__ __device__ inline int bal_block_of(int64_t u, int P, int64_t U) { return (int)(((u + 1) * P - 1) / U); }   // owner of unit u
// workgroups that take part in the remainder (the tiles beyond (tiles / P) P): each of them gets at least one stage
__host__ __device__ inline int bal_rem_blocks(int tiles, int S, int P)
{
    const int64_t Ur = (int64_t)(tiles - (tiles / P) * P) * S;
    return Ur < P ? (int)Ur : P;
}
// slabs of tile t written by a balanced launch of P workgroups over tiles x S stages: whole tiles have one, a
// remainder tile one per workgroup that shared its stages
__host__ __device__ inline int bal_tile_slabs(int t, int tiles, int S, int P)
{
    const int Tf = (tiles / P) * P;
    if (t < Tf) return 1;
    const int64_t Ur = (int64_t)(tiles - Tf) * S, u0 = (int64_t)(t - Tf) * S;
    const int Pr = bal_rem_blocks(tiles, S, P);
    return bal_block_of(u0 + S - 1, Pr, Ur) - bal_block_of(u0, Pr, Ur) + 1;
}
hipError_t launch_gemm_planes(int la, int lb, const PlaneGemmArgs& g, hipStream_t s);
hipError_t launch_gemm_planes_bal(int la, int lb, const PlaneGemmArgs& g, hipStream_t s);
int bal_max_segments(int tiles, int stages, int P);     // slabs a balanced fused == 0 launch needs
int bal_segment_host(int tiles, int S, int P, int w, int k, int* tile, int* s0, int* s1, int* nseg, int* nt);
hipError_t launch_split_planes(const float* X, int64_t rows, int64_t ld, unsigned short* P, int64_t plane_stride, hipStream_t s);
hipError_t launch_gather_planes(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src, const void* idx, int idx64,
                                int64_t n_idx, float* dst, int64_t ld_dst, unsigned short* P, int64_t plane_stride, hipStream_t s);

// blocks (= cost partials) the activation epilogue launches for a [rows, ld] output
int epilogue_blocks(int64_t rows, int64_t ld);
int epilogue_cw(int64_t rows, int64_t ld);
void set_epilogue_cw(int cw);
void set_epilogue_threads(int t);
inline int row_groups(int64_t B) { return (int)((B + 3) / 4); }

hipError_t launch_gemm(int la, int lb, const GemmArgs& g, hipStream_t s);
hipError_t launch_act_epilogue(const EpiArgs& e, hipStream_t s);
hipError_t launch_sum_slabs(const float* slabs, int nsplit, int64_t slab_stride, int64_t n,
                            float* out, hipStream_t s);
hipError_t launch_gather_slim(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src, const void* idx, int idx64,
                              int64_t n_idx, float* dst, int64_t ld_dst, int blocks, int threads, hipStream_t s);
hipError_t launch_gather(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src,
                         const void* idx, int idx64, int64_t n_idx, float* dst, int64_t ld_dst,
                         hipStream_t s);
hipError_t launch_colsum_groups(const float* X, const float* Y, int rows, int64_t ld, float* out, hipStream_t s);
FinArgs make_fin_args(const float* posP, const float* negP, const float* partV, int ngroups, int64_t ldh, int64_t ldv,
                      const float* cost_partials, int n_cost, float* s_h, float* s_v, float* cost,
                      const BiasUpd* bias_update);
hipError_t launch_finalize_stats(const float* posP, const float* negP, const float* partV, int ngroups,
                                 int64_t ldh, int64_t ldv, const float* cost_partials, int n_cost,
                                 float* s_h, float* s_v, float* cost, const BiasUpd* bias_update, hipStream_t s);
// slabs != NULL (phase 0 only): the S block is read as the sum of `nslab` split-K slabs
hipError_t launch_update(const mdbn_update_args& a, hipStream_t s, const float* slabs = nullptr, int nslab = 1,
                         int64_t slab_stride = 0, unsigned short* Wp = nullptr);
hipError_t launch_free_energy(const float* slabs, int nsplit, int64_t slab_stride, int64_t ldh, int H,
                              const float* hbias, const float* x, int64_t ldv, int V, const float* vbias,
                              int gauss, int64_t rows, float* out, hipStream_t s);
hipError_t launch_round_flip(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, int64_t flip_col, hipStream_t s);
hipError_t launch_pl_cost(const float* fe, const float* fe_flip, int64_t rows, float n_visible, float* out, hipStream_t s);
hipError_t launch_recon_cost(const float* pre, int64_t ldp, const float* tgt, int64_t ldt, int64_t rows, int64_t cols, int gauss,
                             float scale, float* partials, int n_partials, float* out, hipStream_t s);
hipError_t launch_tanh(float* x, int64_t rows, int64_t cols, int64_t ld, hipStream_t s);
hipError_t launch_count_nonfinite(const float* x, int64_t n, int* count, hipStream_t s);
hipError_t launch_narrow_bf16(const float* x, unsigned short* y, int64_t n, hipStream_t s);
hipError_t launch_widen_bf16(const unsigned short* y, float* x, int64_t n, hipStream_t s);
hipError_t launch_rng_fill(float* out, int64_t rows, int64_t cols, int64_t ld, const PhiloxKey& k,
                           int normal, hipStream_t s);

}  // namespace mdbn
