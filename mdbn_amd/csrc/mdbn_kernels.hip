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
#include <stdint.h>
#include "philox.h"
#include "mdbn_kernels.h"

namespace mdbn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int LDK = BK + 1;               // row stride of a K-contiguous LDS tile [128][33]
constexpr int TILE_FLOATS = 128 * LDK;    // >= 32*128 (MN-contiguous tile [32][128])
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * (int)sizeof(float);   // A,B x 2 buffers

// ----------------------------------------------------------------------------------
// operand staging: global -> registers -> LDS
// ----------------------------------------------------------------------------------
template <int LAY>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t ld, int MN, int K,
                                          int mn0, int k0, float4 (&r)[4])
{
    const int tid = threadIdx.x;
    if (LAY == LAY_K) {
        // P[mn][k], k contiguous: 8 threads cover one 32-float row slice, 32 rows per pass
        const int c = tid & 7, rr = tid >> 3;
        const int kq = k0 + 4 * c;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = mn0 + rr + 32 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < MN) {
                const float* src = P + (int64_t)row * ld + kq;
                if (kq + 3 < K) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (kq + 0 < K) v.x = src[0];
                    if (kq + 1 < K) v.y = src[1];
                    if (kq + 2 < K) v.z = src[2];
                }
            }
            r[p] = v;
        }
    } else {
        // P[k][mn], mn contiguous: 32 threads cover one 128-float row, 8 k-rows per pass
        const int c = tid & 31, rr = tid >> 5;
        const int mq = mn0 + 4 * c;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int krow = k0 + rr + 8 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (krow < K) {
                const float* src = P + (int64_t)krow * ld + mq;
                if (mq + 3 < MN) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (mq + 0 < MN) v.x = src[0];
                    if (mq + 1 < MN) v.y = src[1];
                    if (mq + 2 < MN) v.z = src[2];
                }
            }
            r[p] = v;
        }
    }
}

template <int LAY>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&r)[4])
{
    const int tid = threadIdx.x;
    if (LAY == LAY_K) {
        const int c = tid & 7, rr = tid >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* d = T + (rr + 32 * p) * LDK + 4 * c;     // odd stride: reads conflict-free
            d[0] = r[p].x; d[1] = r[p].y; d[2] = r[p].z; d[3] = r[p].w;
        }
    } else {
        const int c = tid & 31, rr = tid >> 5;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(T + (rr + 8 * p) * 128 + 4 * c) = r[p];
    }
}

// MFMA operand of lane (i = lane & 31, h = lane >> 5): element [mn = base + i][k = kk + h]
template <int LAY>
__device__ __forceinline__ float frag(const float* __restrict__ T, int mn, int k)
{
    return LAY == LAY_K ? T[mn * LDK + k] : T[k * 128 + mn];
}

// ----------------------------------------------------------------------------------
// C[ks] = A[:, kchunk ks] * B[kchunk ks, :]   (split-K slabs, no epilogue)
// Replaces tensor.dot at rbm.py:168,198,226,411-412,650,685 and mlp.py:103.
// ----------------------------------------------------------------------------------
template <int LA, int LB>
__global__ __launch_bounds__(NTHREADS) void gemm_splitk_kernel(GemmArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // buffer b: A tile at smem + 2*b*TILE_FLOATS, B tile right behind it

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
    const int nt = (kend - kbeg + BK - 1) / BK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    float4 ra[4], rb[4];
    if (nt > 0) {
        load_tile<LA>(g.A, g.lda, g.M, kend, m0, kbeg, ra);
        load_tile<LB>(g.B, g.ldb, g.N, kend, n0, kbeg, rb);
        store_tile<LA>(smem, ra);
        store_tile<LB>(smem + TILE_FLOATS, rb);
    }
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int cur = it & 1;
        if (it + 1 < nt) {                      // prefetch next slice into registers
            load_tile<LA>(g.A, g.lda, g.M, kend, m0, kbeg + (it + 1) * BK, ra);
            load_tile<LB>(g.B, g.ldb, g.N, kend, n0, kbeg + (it + 1) * BK, rb);
        }
        const float* at = smem + cur * (2 * TILE_FLOATS);
        const float* bt = at + TILE_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = frag<LA>(at, wm + i, kk + h);
            const float a1 = frag<LA>(at, wm + 32 + i, kk + h);
            const float b0 = frag<LB>(bt, wn + i, kk + h);
            const float b1 = frag<LB>(bt, wn + 32 + i, kk + h);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (it + 1 < nt) {                      // other buffer: last read one barrier ago
            float* nx = smem + (cur ^ 1) * (2 * TILE_FLOATS);
            store_tile<LA>(nx, ra);
            store_tile<LB>(nx + TILE_FLOATS, rb);
        }
        __syncthreads();
    }

    // accumulator (32x32): col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn + 32 * b + i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < g.M && col < g.Nst) C[(int64_t)row * g.ldc + col] = acc[a][b][e];
            }
        }
}

template <int LA, int LB>
static hipError_t launch_gemm_t(const GemmArgs& g, hipStream_t s)
{
    static bool attr_set = false;
    auto kern = gemm_splitk_kernel<LA, LB>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           GEMM_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = g.tiles_m * g.tiles_n * g.splitk;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), GEMM_LDS_BYTES, s, g);
    return hipGetLastError();
}

hipError_t launch_gemm(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (la == LAY_K && lb == LAY_MN) return launch_gemm_t<LAY_K, LAY_MN>(g, s);
    if (la == LAY_K && lb == LAY_K) return launch_gemm_t<LAY_K, LAY_K>(g, s);
    if (la == LAY_MN && lb == LAY_MN) return launch_gemm_t<LAY_MN, LAY_MN>(g, s);
    return hipErrorInvalidValue;
}

// ----------------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x)
{
    return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x));
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t s = ph + r;
            w[r] = s < 4 ? lo[s & 3] : hi[s & 3];
        }
    }
}

__device__ __forceinline__ float comp(const float4& v, int j)
{
    return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}
__device__ __forceinline__ void setc(float4& v, int j, float x)
{
    if (j == 0) v.x = x; else if (j == 1) v.y = x; else if (j == 2) v.z = x; else v.w = x;
}

// ----------------------------------------------------------------------------------
// split-K reduce + bias + activation + sampling epilogue.
//   gauss = 0: pre = sum + bias ; mean = sigmoid(pre) ; sample = (u < mean)
//              (propup/sample_h_given_v rbm.py:198-213, propdown/sample_v_given_h :226-240)
//   gauss = 1: mean = pre = sum + bias ; sample = mean + N(0,1)      (GRBM rbm.py:650-658)
//   cost (target != NULL): sum of BCE(sigmoid(pre), target) (rbm.py:479-480) or of
//              (sigmoid(mean) - target)^2 (rbm.py:697), one partial per block.
// One thread = 4 rows x 4 columns: float4 traffic, one Philox block per column.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_epilogue_kernel(EpiArgs e)
{
    __shared__ float red[4];
    const int ld4 = (int)(e.ld >> 2);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int rg = (int)(idx / ld4), cq = (int)(idx - (int64_t)rg * ld4);
    const int r0 = rg * 4, c0 = cq * 4;
    float cost = 0.f;
    if (r0 < e.rows) {
        const float4 bias4 = make_float4(c0 + 0 < e.cols ? e.bias[c0 + 0] : 0.f,
                                         c0 + 1 < e.cols ? e.bias[c0 + 1] : 0.f,
                                         c0 + 2 < e.cols ? e.bias[c0 + 2] : 0.f,
                                         c0 + 3 < e.cols ? e.bias[c0 + 3] : 0.f);
        float4 pre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < e.rows) {
                const float* p = e.slabs + (int64_t)(r0 + r) * e.ld + c0;
                for (int s = 0; s < e.nsplit; ++s) {
                    const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)s * e.slab_stride);
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
                a.x += bias4.x; a.y += bias4.y; a.z += bias4.z; a.w += bias4.w;
            }
            pre[r] = a;
        }
        uint32_t wa[4][4], wb[4][4];       // [col][row]
        const bool need_u = e.sample != nullptr;
        const bool need_z = need_u && e.gauss;
        if (need_u) {
            const uint64_t g0 = e.rng.row_offset + (uint64_t)r0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                philox_rows4(e.rng, e.rng.draw, g0, (uint32_t)(c0 + j), wa[j]);
                if (need_z) philox_rows4(e.rng, e.rng.draw | MDBN_NORMAL_BIT, g0, (uint32_t)(c0 + j), wb[j]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r0 + r >= e.rows) continue;
            const int64_t off = (int64_t)(r0 + r) * e.ld + c0;
            float4 mean4, samp4, tgt4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e.target) tgt4 = *reinterpret_cast<const float4*>(e.target + (int64_t)(r0 + r) * e.ld_target + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool live = c0 + j < e.cols;
                const float x = comp(pre[r], j);
                float m, sv = 0.f;
                if (e.gauss) {
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
                    const float tg = comp(tgt4, j);
                    if (e.gauss) { const float d = sigmoidf_(x) - tg; cost += d * d; }
                    else cost += tg * softplusf_(-x) + (1.0f - tg) * softplusf_(x);
                }
                if (!live) { m = 0.f; sv = 0.f; setc(pre[r], j, 0.f); }   // keep pad columns zero
                setc(mean4, j, m * e.mean_scale);
                setc(samp4, j, sv);
            }
            if (e.pre) *reinterpret_cast<float4*>(e.pre + off) = pre[r];
            if (e.mean) *reinterpret_cast<float4*>(e.mean + off) = mean4;
            if (e.sample) *reinterpret_cast<float4*>(e.sample + off) = samp4;
        }
    }
    if (e.cost_partials) {
        const float tot = block_sum(cost, red);
        if (threadIdx.x == 0) e.cost_partials[blockIdx.x] = tot;
    }
}

hipError_t launch_act_epilogue(const EpiArgs& e, hipStream_t s)
{
    hipLaunchKernelGGL(act_epilogue_kernel, dim3(epilogue_blocks(e.rows, e.ld)), dim3(256), 0, s, e);
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
        for (int s = 0; s < nsplit; ++s) {
            const float4 v = reinterpret_cast<const float4*>(slabs + (int64_t)s * slab_stride)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
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
    const int64_t r = blockIdx.x;
    int64_t s = r;
    if (idx) s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
    if (s < 0) s += n_rows;                       // numpy-style negative index
    s = s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);   // never fault on a bad index
    const float4* in = reinterpret_cast<const float4*>(src + s * ld_src);
    float4* out = reinterpret_cast<float4*>(dst + r * ld_dst);
    for (int64_t c = threadIdx.x; c < ld4; c += blockDim.x) out[c] = in[c];
}

hipError_t launch_gather(const float* src, int64_t n_rows, int64_t cols_ld, int64_t ld_src,
                         const void* idx, int idx64, int64_t n_idx, float* dst, int64_t ld_dst,
                         hipStream_t s)
{
    if (n_idx <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_idx), dim3(256), 0, s, src, n_rows, ld_src,
                       idx, idx64, cols_ld >> 2, dst, ld_dst);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// column sums for the bias statistics (rbm.py:416-417), two deterministic passes.
// Pass 1: partial[c][col] = sum of the rows of chunk c (64 rows, never straddling B).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int B, int64_t ld,
                                                             int nch, float* __restrict__ partial)
{
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y, half = c / nch, cc = c - half * nch;
    const int rbeg = half * B + cc * 64, rend = min(half * B + B, rbeg + 64);
    const int64_t col = ((int64_t)blockIdx.x * 64 + lane) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < ld)
        for (int r = rbeg + wave; r < rend; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)r * ld + col);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    red[wave][lane] = a;
    __syncthreads();
    if (wave == 0 && col < ld) {
        float4 t = red[0][lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) { t.x += red[k][lane].x; t.y += red[k][lane].y; t.z += red[k][lane].z; t.w += red[k][lane].w; }
        *reinterpret_cast<float4*>(partial + (int64_t)c * ld + col) = t;
    }
}

hipError_t launch_colsum_partial(const float* X, int B, int64_t ld, float* partial, hipStream_t s)
{
    const int nch = colsum_chunks(B);
    dim3 grid((unsigned)((ld / 4 + 63) / 64), (unsigned)(2 * nch));
    hipLaunchKernelGGL(colsum_partial_kernel, grid, dim3(256), 0, s, X, B, ld, nch, partial);
    return hipGetLastError();
}

// Pass 2: s_h = sum_c partP[c]  (P2's second half already holds -nh_mean)
//         s_v = sum_{c<nch} partV[c] - sum_{c>=nch} partV[c] ; cost_sum = sum cost partials
__global__ __launch_bounds__(256) void finalize_stats_kernel(const float* __restrict__ partP, const float* __restrict__ partV,
                                                             int nch, int64_t ldh, int64_t ldv,
                                                             const float* __restrict__ cost_partials, int n_cost,
                                                             float* __restrict__ s_h, float* __restrict__ s_v,
                                                             float* __restrict__ cost)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ldh) {
        float a = 0.f;
        for (int c = 0; c < 2 * nch; ++c) a += partP[(int64_t)c * ldh + i];
        s_h[i] = a;
    } else if (i < ldh + ldv) {
        const int64_t j = i - ldh;
        float p = 0.f, n = 0.f;
        for (int c = 0; c < nch; ++c) p += partV[(int64_t)c * ldv + j];
        for (int c = nch; c < 2 * nch; ++c) n += partV[(int64_t)c * ldv + j];
        s_v[j] = p - n;
    }
    if (blockIdx.x == gridDim.x - 1 && cost_partials) {     // last block also totals the cost
        __shared__ float red[4];
        float a = 0.f;
        for (int k = threadIdx.x; k < n_cost; k += blockDim.x) a += cost_partials[k];
        const float t = block_sum(a, red);
        if (threadIdx.x == 0) { cost[0] = t; cost[1] = 0.f; cost[2] = 0.f; cost[3] = 0.f; }
    }
}

hipError_t launch_finalize_stats(const float* partP, const float* partV, int B, int64_t ldh, int64_t ldv,
                                 const float* cost_partials, int n_cost, float* s_h, float* s_v,
                                 float* cost, hipStream_t s)
{
    const int grid = (int)((ldh + ldv + 255) / 256) + 1;
    hipLaunchKernelGGL(finalize_stats_kernel, dim3(grid), dim3(256), 0, s, partP, partV, colsum_chunks(B),
                       ldh, ldv, cost_partials, n_cost, s_h, s_v, cost);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// parameter update (rbm.py:347-365): lambda shrink, EMA "speed", lagged apply
//   g   = (S / batch_size - weightcost * W0) / (1 + 2 lr l1 / (|W| + eps))
//   W'  = W * (1 - 2 lr l2) / (1 + 2 lr l1 / (|W| + eps)) + W_speed(old) * lr
//   Ws' = g + (W_speed - g) * momentum
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_W_kernel(float4* __restrict__ W, float4* __restrict__ Ws,
                                                       const float4* __restrict__ W0, const float4* __restrict__ S,
                                                       int64_t n4, float lr, float l1, float l2, float wc,
                                                       float mu, float inv_bs)
{
    const float two_lr_l1 = 2.0f * lr * l1;
    const float decay = 1.0f - 2.0f * lr * l2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 w = W[i], sp = Ws[i], st = S[i];
        const float4 wc0 = W0 ? W0[i] : w;
        float4 wn, sn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wj = comp(w, j), spj = comp(sp, j);
            float g = comp(st, j) * inv_bs - wc * comp(wc0, j);
            float m = decay;
            if (l1 != 0.0f) {
                const float shrink = 1.0f + two_lr_l1 / (fabsf(wj) + 0.001f);
                g = g / shrink;
                m = decay / shrink;
            }
            setc(sn, j, g + (spj - g) * mu);
            setc(wn, j, wj * m + spj * lr);
        }
        W[i] = wn;
        Ws[i] = sn;
    }
}

__global__ __launch_bounds__(256) void update_bias_kernel(float* __restrict__ hb, float* __restrict__ hbs,
                                                          const float* __restrict__ s_h, int64_t H,
                                                          float* __restrict__ vb, float* __restrict__ vbs,
                                                          const float* __restrict__ s_v, int64_t V,
                                                          float lr, float mu, float inv_rows,
                                                          const float* __restrict__ cost_sum, float cost_scale,
                                                          float* __restrict__ cost_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && cost_out) cost_out[0] = cost_sum[0] * cost_scale;
    if (i < H) {
        const float g = s_h[i] * inv_rows, sp = hbs[i];
        hbs[i] = g + (sp - g) * mu;
        hb[i] = hb[i] + sp * lr;
    } else if (i < H + V) {
        const int64_t j = i - H;
        const float g = s_v[j] * inv_rows, sp = vbs[j];
        vbs[j] = g + (sp - g) * mu;
        vb[j] = vb[j] + sp * lr;
    }
}

hipError_t launch_update(const mdbn_update_args& a, hipStream_t s)
{
    const int64_t n4 = (a.V * a.ldh) >> 2;
    const float* S = a.stats;
    const float* s_h = a.stats + a.V * a.ldh;
    const float* s_v = s_h + a.ldh;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(update_W_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<float4*>(a.W),
                       reinterpret_cast<float4*>(a.W_speed), reinterpret_cast<const float4*>(a.W0),
                       reinterpret_cast<const float4*>(S), n4, a.lr, a.lambda_1, a.lambda_2, a.weightcost,
                       a.momentum, 1.0f / a.batch_size);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int gb = (int)((a.H + a.V + 255) / 256);
    hipLaunchKernelGGL(update_bias_kernel, dim3(gb), dim3(256), 0, s, a.hbias, a.hbias_speed, s_h, a.H,
                       a.vbias, a.vbias_speed, s_v, a.V, a.lr, a.momentum, 1.0f / a.n_rows,
                       s_v + a.ldv, a.cost_scale, a.cost_out);
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
        float a = hbias[j];
        for (int s = 0; s < nsplit; ++s) a += slabs[(int64_t)s * slab_stride + r * ldh + j];
        acc -= (double)softplusf_(a);
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

}  // namespace mdbn
