// Group-chain CD-k: the positive phase and the whole Gibbs chain of a mid-size layer at B > 32 in ONE launch (the layers
// of BASELINE configs 4 / 5 whose W misses one CU's LDS: 256 -> 200, 1024 -> 256; reference shapes AMLsm2.py:242-340,
// MDBN.py:31-35, the scan of rbm.py:318-336).  On the multi-launch path such a step is 2k + 1 GEMM launches (+ epilogue
// launches) of 8-17 us each, all of it launch / first-touch / split-K-slab overhead.
//
// The minibatch is cut into SLABS of 32 rows (one M tile of v_mfma_f32_32x32x16_bf16); a slab's chain runs on a GROUP of g
// workgroups (g = 2, 4 or 8), member m holding rows [m Vb, (m + 1) Vb) of W in LDS as float32 for the WHOLE launch.  With
// whole rows of W a member's slice of every visible pass (v = act(h W^T + vbias)) is complete inside the member; only the
// upward product needs the other members: each writes its [32, H] partial, and ONE exchange per pass gives every member
// the sum -- all members then apply the hidden activation to the identical sum (same Philox words), so the chain state
// never has to travel.  Arithmetic: the thin-batch kernels' (mdbn_thin.hip): fragments read from the float32 LDS image,
// split exactly into three bf16 pieces in registers (mdbn_bf16x3.h), six / three piece products.
//
// The exchange is the one inter-workgroup hand-off of the library that stayed inside a launch (four earlier ones lost to
// kernel boundaries: here one launch replaces 2k + 1).  Protocol (MI355X guide, Guideline 16, R1 form): payload written
// with agent-scope relaxed atomic stores (write-through), every storing wave drains (s_waitcnt vmcnt(0)), workgroup
// barrier, ONE lane stores the member's flag = the exchange's sequence number (unique per launch and exchange: the flag
// words belong to the context and only this protocol writes them); the g lowest lanes poll the g flags with agent-scope
// relaxed loads (bounded: a partner that never arrives sets the launch's error word instead of hanging the chip), barrier,
// then every load of the payload is an agent-scope relaxed atomic load.  Payload and flags alternate between two buffers;
// a member can only reach exchange e + 2 after every partner has left exchange e.  Placement-independent: residency comes
// from the grid (groups x g <= CUs, one workgroup per CU by its LDS footprint).
//
// Outputs exactly as the multi-launch path leaves them for the statistics GEMM: V2 = [v0; nv], P2 = [ph; -nh], the
// 4-row column partials of the bias statistics, one cost partial per (slab, member), hs / vs and the chain taps.
#include <hip/hip_runtime.h>
#include "mdbn_gchain.h"
#include "mdbn_device.h"
#include "mdbn_bf16x3.h"

namespace mdbn {

extern __shared__ __align__(16) float gc_smem[];

__device__ __forceinline__ int64_t gc_src_row(const void* idx, int idx64, int64_t r, int64_t n_rows)
{
    if (!idx) return r;
    int64_t s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
    if (s < 0) s += n_rows;
    return s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);
}

__device__ __forceinline__ void gc_store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float gc_load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NT2 = 32-column tiles of the upward product per wave (ldh <= 256 NT2); GAUSS: Gaussian visible units (linear mean, the chain
// goes on from the mean: rbm.py:647-671), else Bernoulli (sigmoid + sample: rbm.py:226-248)
template <int NT2, bool GAUSS>
__global__ __launch_bounds__(GC_NT) void gchain_kernel(GChainArgs a)
{
    constexpr int XPV = GAUSS ? 3 : 1;      // pieces of the visible operand of the upward product inside the chain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, kh = lane >> 5;
    const int g = a.g, m = blockIdx.x % g, sg = blockIdx.x / g;
    const int PW = a.PW;
    const int q4 = (int)(a.ldh >> 2);
    const int K16 = ((int)a.ldh + 15) & ~15;
    const int PH = K16 + 8;
    const int r0 = m * a.Vb, r1 = min(a.V, r0 + a.Vb), nrows = max(r1 - r0, 0);
    const int R16 = a.Vb, R32 = (a.Vb + 31) & ~31, PX = thin_pitch(R16);
    const int ntile1 = R32 >> 5, S1 = a.S1;

    float* Wf = gc_smem;
    unsigned short* hKb = reinterpret_cast<unsigned short*>(gc_smem + R16 * PW);
    float* red = reinterpret_cast<float*>(hKb);
    const int hk_bytes = 32 * PH * 2, red_bytes = S1 * 32 * R32 * 4;
    // the visible operand of the upward product, float32 [32][PX] (split into bf16 pieces on the way into the MFMA, like W)
    float* xF = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(hKb) + ((max(hk_bytes, red_bytes) + 15) & ~15));
    float* hbL = xF + 32 * PX;              // hbias [K16] and this member's vbias [R16]: read in every epilogue of every pass
    float* vbL = hbL + K16;
    for (int e = tid; e < K16; e += GC_NT) hbL[e] = e < a.H ? a.hbias[e] : 0.f;
    for (int e = tid; e < R16; e += GC_NT) vbL[e] = e < nrows ? a.vbias[r0 + e] : 0.f;

    // ---- stage: this member's rows of W -> Wf, once for the whole launch (pad rows / pad columns: zeros)
    {
        const int k4 = K16 >> 2;
        for (int rb = wave; rb < R16; rb += 64) {
            float4 v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = rb + 8 * u;
                const float* src = a.W + (int64_t)(r0 + min(row, max(nrows - 1, 0))) * a.ldh;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int c4 = lane + 64 * c;
                    v[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < nrows && 64 * c < q4) {
                        v[u][c] = *reinterpret_cast<const float4*>(src + 4 * min(c4, q4 - 1));
                        if (c4 >= q4) v[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = rb + 8 * u;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int c4 = lane + 64 * c;
                    if (row < R16 && c4 < k4) *reinterpret_cast<float4*>(Wf + row * PW + 4 * c4) = v[u][c];
                }
            }
        }
    }

    float* xb = a.xbuf + (int64_t)sg * 2 * g * 32 * a.ldh;
    unsigned* fl = a.flags + sg * 2 * g;
    unsigned ex = 0;                        // exchanges this group has made in this launch
    int jt[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) jt[t] = min(32 * (wave + 8 * t) + ln, (int)a.ldh - 1);

    for (int slab = sg; slab < a.nslab; slab += a.nsg) {
        const int b0 = 32 * slab, nb = min(32, a.B - b0);
        float cost = 0.f;

        // ---- x = train_set_x[indexes][b0 .. b0 + nb) over this member's columns -> V2 (v0) and the LDS tile xF
        __syncthreads();                    // (the previous slab's readers of xF are done)
        for (int e = tid; e < 32 * R16; e += GC_NT) {
            const int b = e / R16, i = e - b * R16;
            float xv = 0.f;
            if (b < nb && i < nrows) {
                xv = a.data[gc_src_row(a.idx, a.idx64, b0 + b, a.n_data) * a.ld_data + r0 + i];
                a.V2[(int64_t)(b0 + b) * a.ldv + r0 + i] = xv;
            }
            xF[b * PX + i] = xv;
        }
        if (m == g - 1)                     // pad columns of the gathered rows stay zero
            for (int e = tid; e < nb * (int)(a.ldv - a.V); e += GC_NT) {
                const int b = e / (int)(a.ldv - a.V), c = e - b * (int)(a.ldv - a.V);
                a.V2[(int64_t)(b0 + b) * a.ldv + a.V + c] = 0.f;
            }
        __syncthreads();

        for (int t = 0; t <= a.k; ++t) {
            const bool last = t == a.k;
            if (t > 0) {
                // ---- visible pass of Gibbs step t: v_pre[b][i] = sum_j h[b][j] W[r0 + i][j]   (complete inside the member)
                f32x16 acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
                const int t1 = wave % ntile1, ks = wave / ntile1;
                const bool has1 = ks < S1 && wave < ntile1 * S1;
                if (has1) {
                    const int nsteps = K16 >> 4, nper = (nsteps + S1 - 1) / S1;
                    const int s_end = min(nsteps, (ks + 1) * nper);
                    const int rowB = min(32 * t1 + ln, R16 - 1);
                    for (int s = ks * nper; s < s_end; ++s) {
                        const tu32x4 aw = *reinterpret_cast<const tu32x4*>(hKb + ln * PH + 16 * s + 8 * kh);
                        const float4 w0 = *reinterpret_cast<const float4*>(Wf + rowB * PW + 16 * s + 8 * kh);
                        const float4 w1 = *reinterpret_cast<const float4*>(Wf + rowB * PW + 16 * s + 8 * kh + 4);
                        const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                        tbf16x8 fb[3], fa[3];
                        th_split8(f, fb);
                        fa[0] = __builtin_bit_cast(tbf16x8, aw); fa[1] = fa[0]; fa[2] = fa[0];
                        th_mma<1>(acc1, fa, fb);
                    }
                }
                __syncthreads();            // every wave is done reading hKb: red may overwrite it
                if (has1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
                        red[(ks * 32 + b) * R32 + 32 * t1 + ln] = acc1[r];
                    }
                }
                __syncthreads();
                // visible activation: thread = 4 rows x 1 column (one Philox block)
                for (int e = tid; e < R32 * 8; e += GC_NT) {
                    const int i = e % R32, bq = e / R32;
                    const int col = r0 + i;
                    const bool live = i < nrows;
                    const float vb_e = i < R16 ? vbL[i] : 0.f;
                    float tg_e[4] = {0.f, 0.f, 0.f, 0.f};
                    if (live && last) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * bq + r < nb) tg_e[r] = a.V2[(int64_t)(b0 + 4 * bq + r) * a.ldv + col];
                    }
                    float x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sacc = 0.f;
                        for (int w = 0; w < S1; ++w) sacc += red[(w * 32 + 4 * bq + r) * R32 + i];
                        x[r] = sacc + vb_e;
                    }
                    uint32_t wa[4] = {0u, 0u, 0u, 0u};
                    if (!GAUSS) philox_rows4(a.rng, (uint32_t)(2 * t - 1), a.rng.row_offset + (uint64_t)(b0 + 4 * bq), (uint32_t)col, wa);
                    float csum = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int b = 4 * bq + r;
                        const bool on = live && b < nb;
                        float mval, sv;
                        if (GAUSS) { mval = x[r]; sv = mval; }
                        else { mval = sigmoidf_(x[r]); sv = philox_u01(wa[r]) < mval ? 1.0f : 0.0f; }
                        if (on && last) {
                            if (GAUSS) { const float d = sigmoidf_(x[r]) - tg_e[r]; cost += d * d; }           // rbm.py:697
                            else cost += tg_e[r] * softplusf_(-x[r]) + (1.0f - tg_e[r]) * softplusf_(x[r]);       // rbm.py:479-480
                            csum += tg_e[r] - mval;                                                              // rbm.py:417
                        }
                        if (on) {
                            if (last) a.V2[(int64_t)(a.B + b0 + b) * a.ldv + col] = mval;
                            if (!GAUSS) {
                                a.vs[(int64_t)(b0 + b) * a.ldv + col] = sv;
                                if (a.trace_v) a.trace_v[((int64_t)(t - 1) * a.B + b0 + b) * a.ldv + col] = sv;
                            }
                        }
                        if (i < R16) xF[b * PX + i] = on ? sv : 0.f;
                    }
                    if (last && live && 4 * bq < nb) a.colV[(int64_t)((b0 >> 2) + bq) * a.ldv + col] = csum;
                }
                if (last && m == g - 1)     // pad columns of nv and of its column partials
                    for (int e = tid; e < 32 * (int)(a.ldv - a.V); e += GC_NT) {
                        const int b = e / (int)(a.ldv - a.V), c = e - b * (int)(a.ldv - a.V);
                        if (b < nb) a.V2[(int64_t)(a.B + b0 + b) * a.ldv + a.V + c] = 0.f;
                        if ((b & 3) == 0 && b < nb) a.colV[(int64_t)((b0 + b) >> 2) * a.ldv + a.V + c] = 0.f;
                    }
                __syncthreads();
            }

            // ---- upward product of this member's rows: partial[b][j] = sum_i x[b][i] W[r0 + i][j]
            f32x16 acc2[NT2];
#pragma unroll
            for (int u = 0; u < NT2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[u][r] = 0.f;
            if (32 * wave < (int)a.ldh) {
                const int nst2 = (nrows + 15) >> 4;
                for (int s = 0; s < nst2; ++s) {
                    tbf16x8 fa[3];
                    {
                        const float4 x0 = *reinterpret_cast<const float4*>(xF + ln * PX + 16 * s + 8 * kh);
                        const float4 x1 = *reinterpret_cast<const float4*>(xF + ln * PX + 16 * s + 8 * kh + 4);
                        const float fx[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        if (t == 0 || XPV == 3) th_split8(fx, fa);
                        else {              // 0/1 samples: the upper halves ARE the bf16 values
                            tu32x4 q;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                q[e] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, fx[2 * e + 1]), __builtin_bit_cast(unsigned, fx[2 * e]), 0x07060302u);
                            fa[0] = __builtin_bit_cast(tbf16x8, q); fa[1] = fa[0]; fa[2] = fa[0];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NT2; ++u) {
                        if (32 * (wave + 8 * u) < (int)a.ldh) {
                            const float* wp = Wf + (16 * s + 8 * kh) * PW + jt[u];
                            float f[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = wp[e * PW];
                            tbf16x8 fb[3];
                            th_split8(f, fb);
                            if (t == 0 || XPV == 3) th_mma<3>(acc2[u], fa, fb);
                            else th_mma<1>(acc2[u], fa, fb);
                        }
                    }
                }
            }

            // ---- exchange: publish the partial, wait for the g members, then sum all g in member order (identical everywhere)
            const unsigned seq = a.seq0 + ex;
            const int par = (int)(ex & 1u);
            ++ex;
            float* mine = xb + (int64_t)(par * g + m) * 32 * a.ldh;
#pragma unroll
            for (int u = 0; u < NT2; ++u) {
                const int j = 32 * (wave + 8 * u) + ln;
                if (j < (int)a.ldh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
                        gc_store_sc1(mine + (int64_t)b * a.ldh + j, acc2[u][r]);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fl + par * g + m, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < g) {
                int spins = 0;
                while (__hip_atomic_load(fl + par * g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > GC_SPIN_LIMIT) { atomicAdd(a.error, 1u); break; }        // a partner that never arrives: no hang
                }
            }
            __syncthreads();

            // ---- hidden activation on the sum: item = (4-row block, column); every member computes every item (the chain
            // state is needed whole), the global outputs of a column go out from ONE member
            const bool need_sample = !last;
            const float* pbase = xb + (int64_t)par * g * 32 * a.ldh;
            for (int e = tid; e < 8 * (int)a.ldh; e += GC_NT) {
                const int col = e % (int)a.ldh, bq = e / (int)a.ldh;
                // the item's partials of up to eight members requested together (member by member every member was a round
                // trip to the exchange buffer), summed in member order: identical sums on every member
                float x[4] = {0.f, 0.f, 0.f, 0.f};
                for (int m0 = 0; m0 < g; m0 += 8) {
                    float v[8][4];
#pragma unroll
                    for (int mc = 0; mc < 8; ++mc) {
                        const float* pp = pbase + ((int64_t)min(m0 + mc, g - 1) * 32 + 4 * bq) * a.ldh + col;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[mc][r] = gc_load_sc1(pp + (int64_t)r * a.ldh);
                    }
#pragma unroll
                    for (int mc = 0; mc < 8; ++mc)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] += m0 + mc < g ? v[mc][r] : 0.f;
                }
                const bool live = col < a.H;
                const float hb = hbL[col];
                const bool writer = (col >> 5) % g == m;
                uint32_t wa[4] = {0u, 0u, 0u, 0u};
                if (need_sample) philox_rows4(a.rng, (uint32_t)(2 * t), a.rng.row_offset + (uint64_t)(b0 + 4 * bq), (uint32_t)col, wa);
                float csum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = 4 * bq + r;
                    const bool on = live && b < nb;
                    const float mval = on ? sigmoidf_(x[r] + hb) : 0.f;
                    const float sv = (need_sample && on) ? (philox_u01(wa[r]) < mval ? 1.0f : 0.0f) : 0.f;
                    if (need_sample && col < K16) hKb[b * PH + col] = (unsigned short)(__builtin_bit_cast(unsigned, sv) >> 16);
                    if (writer && b < nb) {
                        const int64_t row = b0 + b;
                        if (t == 0) { a.P2[row * a.ldh + col] = mval; csum += mval; }                      // ph_mean
                        if (last) { a.P2[(a.B + row) * a.ldh + col] = -mval; csum -= (t == 0 ? 0.f : mval); }   // -nh_mean
                        if (need_sample) {
                            a.hs[row * a.ldh + col] = sv;
                            if (a.trace_h) a.trace_h[((int64_t)t * a.B + row) * a.ldh + col] = sv;
                        }
                    }
                }
                if (writer && 4 * bq < nb) {
                    if (t == 0) a.colPpos[(int64_t)((b0 >> 2) + bq) * a.ldh + col] = csum;
                    if (last) a.colPneg[(int64_t)((b0 >> 2) + bq) * a.ldh + col] = (t == 0 ? 0.f : csum);
                }
            }
            if (need_sample)                // columns ldh .. K16 + 7 of the chain-state image: zeros
                for (int e = tid; e < 32 * (PH - (int)a.ldh); e += GC_NT) {
                    const int b = e / (PH - (int)a.ldh), c = e - b * (PH - (int)a.ldh);
                    hKb[b * PH + (int)a.ldh + c] = 0;
                }
            __syncthreads();
        }
        // one cost partial per (slab, member)
        const float tot = block_sum(cost, red);
        if (tid == 0) a.cost_partials[slab * g + m] = tot;
    }
}

template <int NT2, bool GAUSS>
static hipError_t launch_gchain_t(const GChainArgs& a, int lds, hipStream_t s)
{
    auto kern = gchain_kernel<NT2, GAUSS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GC_MAX_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.nsg * a.g), dim3(GC_NT), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_gchain(const GChainArgs& a, int lds, hipStream_t s)
{
    const int nt2 = (a.ldh + 31) / 32 <= 8 ? 1 : 2;
    if (a.gauss) return nt2 == 1 ? launch_gchain_t<1, true>(a, lds, s) : launch_gchain_t<2, true>(a, lds, s);
    return nt2 == 1 ? launch_gchain_t<1, false>(a, lds, s) : launch_gchain_t<2, false>(a, lds, s);
}

}  // namespace mdbn
