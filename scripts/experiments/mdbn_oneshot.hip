// EXPERIMENT (round 5; not built into the library): measured slower than gemm_bf16x6_kernel on every launch it replaced,
// profiles/r05h_oneshot_gemm_ab.log.  Kept as the record of the design; it compiled against csrc/ at commit b859bd2 + the
// `oneshot` flag of GemmArgs.
// Single-shot staged GEMM for SHORT reductions (split-K jobs of <= 256 k per workgroup): the mid-size layers of BASELINE
// configs 4 / 5 at B = 512 (2048 -> 400, 1024 -> 256: reference shapes AMLsm2.py:242-251, MDBN.py:31-35), where the 128 x 128
// tile plan splits K until a workgroup is left with 4-8 slices of 32.  The pipelined kernels (gemm_bf16x6_kernel: producer
// waves, double-buffered 32-deep slices) pay one exposed memory round trip PER SLICE on such jobs -- 13-17 us per launch for
// 2-3 us of arithmetic.  Here a workgroup requests its whole 128 x 128 operand tiles (float32, <= 128 k each: 2 x 64 KB) at
// once -- every load of the workgroup in flight together, ONE round trip -- into LDS images that keep the operands' own
// memory layout, and the eight MFMA waves read fragments straight from those images, splitting each into its three bf16
// pieces in registers on the way into v_mfma_f32_32x32x16_bf16 (mdbn_bf16x3.h; the same six / three piece products and
// order as gemm_bf16x6_kernel: f32-grade results).  A second shot (k 128..255) repeats the stage.  Output: split-K slabs,
// as the kernel it replaces (the activation epilogue / update kernel that follows is unchanged).
//   LAY_K  operand X[rows][ld] (k contiguous): image [128 rows][132], fragment = two ds_read_b128 of one row
//   LAY_MN operand X[k][ld] (rows contiguous): image [128 k][132], fragment = eight ds_read_b32 down one column
#include <hip/hip_runtime.h>
#include "mdbn_kernels.h"
#include "mdbn_device.h"
#include "mdbn_bf16x3.h"

namespace mdbn {

constexpr int OS_NT = 512, OS_KC = 128, OS_P = 132;     // pitch 132: 33 bank quads, odd -> row reads conflict-free
extern __shared__ __align__(16) float os_smem[];

// one 128 x 128 tile of an operand into its LDS image; `r0` = first row (LAY_K) / column (LAY_MN) of the tile, `rows` the
// operand's extent in that direction, k in [k0, kend).  Thread = one float4 column group (tid & 31) of 8 image rows.
template <int LAY>
__device__ __forceinline__ void os_issue(const float* __restrict__ X, int64_t ld, int r0, int rows, int k0, int kend, float4 (&v)[8])
{
    const int c4 = threadIdx.x & 31, rb = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int ir = rb + 16 * u;                       // image row 0..127
        // image row = operand row (LAY_K) or k (LAY_MN); image column = k (LAY_K) or operand row (LAY_MN)
        const int orow = LAY == LAY_K ? r0 + ir : k0 + ir;
        const int ocol = LAY == LAY_K ? k0 + 4 * c4 : r0 + 4 * c4;
        const int rlim = LAY == LAY_K ? rows : kend, clim = LAY == LAY_K ? kend : rows;
        const int orc = min(orow, rlim - 1), occ = min(ocol, (clim - 1) & ~3);        // clamped: always a valid, 16-byte aligned address inside the row's ld
        float4 t = *reinterpret_cast<const float4*>(X + (int64_t)orc * ld + occ);
        if (ocol + 3 >= clim || orow >= rlim) {           // ragged edge / K tail: element-wise zeros
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = ocol + j;
                float val = 0.f;
                if (orow < rlim && cj < clim) val = X[(int64_t)orow * ld + cj];
                e[j] = val;
            }
            t = make_float4(e[0], e[1], e[2], e[3]);
        }
        v[u] = t;
    }
}

__device__ __forceinline__ void os_store(float* img, const float4 (&v)[8])
{
    const int c4 = threadIdx.x & 31, rb = threadIdx.x >> 5;
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(img + (rb + 16 * u) * OS_P + 4 * c4) = v[u];
}

// the eight consecutive-k values of one fragment: element e = k 16 s + 8 h + e of image row / column `rc`
template <int LAY>
__device__ __forceinline__ void os_frag8(const float* img, int rc, int s, int h, float (&f)[8])
{
    if constexpr (LAY == LAY_K) {
        const float4 a = *reinterpret_cast<const float4*>(img + rc * OS_P + 16 * s + 8 * h);
        const float4 b = *reinterpret_cast<const float4*>(img + rc * OS_P + 16 * s + 8 * h + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const float* p = img + (16 * s + 8 * h) * OS_P + rc;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = p[e * OS_P];
    }
}

// AP = pieces of the A operand: 3, or 1 when A holds 0/1 samples (exact in bf16: three products instead of six)
template <int LA, int LB, int AP>
__global__ __launch_bounds__(OS_NT) void gemm_oneshot_kernel(GemmArgs g)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware (split, tile) order, as gemm_bf16x6_kernel: the jobs of one K share and neighbouring tiles meet on one XCD
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qq = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.inner_m) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else           { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);

    float* As = os_smem;
    float* Bs = os_smem + 128 * OS_P;

    float4 va[8], vb[8];
    os_issue<LA>(g.A, g.lda, m0, g.M, kbeg, kend, va);
    os_issue<LB>(g.B, g.ldb, n0, g.N, kbeg, kend, vb);

    if (LA == LAY_MN && g.fin_enabled) {    // statistics GEMM: the finalize units ride on the first round trip
        const int nu = fin_units(g.fin);
        for (int unit = wave * (int)gridDim.x + (int)blockIdx.x; unit <= nu; unit += 8 * (int)gridDim.x) finalize_unit(g.fin, unit, lane);
    }

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const int ln = lane & 31, kh = lane >> 5;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;

    for (int k0 = kbeg; k0 < kend; k0 += OS_KC) {
        os_store(As, va);
        os_store(Bs, vb);
        __syncthreads();
        if (k0 + OS_KC < kend) {            // the next shot travels while this one is multiplied
            os_issue<LA>(g.A, g.lda, m0, g.M, k0 + OS_KC, kend, va);
            os_issue<LB>(g.B, g.ldb, n0, g.N, k0 + OS_KC, kend, vb);
        }
        const int nsteps = (min(OS_KC, kend - k0) + 15) >> 4;
        for (int s = 0; s < nsteps; ++s) {
            float fb8[8];
            os_frag8<LB>(Bs, wn + ln, s, kh, fb8);
            tbf16x8 fb[3];
            th_split8(fb8, fb);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float fa8[8];
                os_frag8<LA>(As, wm + 32 * mt + ln, s, kh, fa8);
                tbf16x8 fa[3];
                if constexpr (AP == 3) th_split8(fa8, fa);
                else {                       // 0/1 samples: the upper halves ARE the bf16 values
                    tu32x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        q[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, fa8[2 * e + 1]), __builtin_bit_cast(unsigned, fa8[2 * e]), 0x07060302u);
                    fa[0] = __builtin_bit_cast(tbf16x8, q); fa[1] = fa[0]; fa[2] = fa[0];
                }
                th_mma<AP>(acc[mt], fa, fb);
            }
        }
        __syncthreads();                    // every wave is done with the images before the next shot overwrites them
    }

    // split-K slab of this job: C[ks][m][n]; columns N .. Nst - 1 receive the exact zeros the zero-filled B tile produced
    float* C = g.C + (int64_t)ks * g.slab_stride;
    const int col = n0 + wn + ln;
    if (col < g.Nst) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < g.M) C[(int64_t)row * g.ldc + col] = acc[mt][r];
            }
    }
}

template <int LA, int LB, int AP>
static hipError_t launch_oneshot_t(const GemmArgs& g, hipStream_t s)
{
    auto kern = gemm_oneshot_kernel<LA, LB, AP>;
    static bool attr_done = false;
    constexpr int lds = 2 * 128 * OS_P * 4;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.splitk), dim3(OS_NT), lds, s, g);
    return hipGetLastError();
}

hipError_t launch_gemm_oneshot(int la, int lb, const GemmArgs& g, hipStream_t s)
{
    if (g.fused || g.kchunk % 16 || g.tiles_m != (g.M + 127) / 128 || g.tiles_n != (g.N + 127) / 128 || (g.lda & 3) || (g.ldb & 3) ||
        g.M < 1 || g.N < 1 || g.K < 1)
        return hipErrorInvalidValue;
    const bool a1 = g.x6 == 2;
    if (la == LAY_K && lb == LAY_MN) return a1 ? launch_oneshot_t<LAY_K, LAY_MN, 1>(g, s) : launch_oneshot_t<LAY_K, LAY_MN, 3>(g, s);
    if (la == LAY_K && lb == LAY_K) return a1 ? launch_oneshot_t<LAY_K, LAY_K, 1>(g, s) : launch_oneshot_t<LAY_K, LAY_K, 3>(g, s);
    if (la == LAY_MN && lb == LAY_MN) return launch_oneshot_t<LAY_MN, LAY_MN, 3>(g, s);
    return hipErrorInvalidValue;
}

}  // namespace mdbn
