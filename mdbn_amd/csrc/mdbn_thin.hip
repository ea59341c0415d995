// Thin-batch CD-k step: minibatches of <= 32 rows (the reference's batch_size = 20, MDBN.py:46 / AMLsm2.py:245) on layers
// whose W does not fit one CU's LDS.  At this batch every product of the step touches each weight once or twice, so the
// step is a STREAM over W and the design goal is to read W as few times as the chain's dependencies allow:
//
//   thin_pass_kernel<0>   positive phase (rbm.py:303): each workgroup owns a contiguous range of W's rows (visible units),
//                         gathers x = train_set_x[indexes] for those columns itself and writes ONE partial of x W.
//   thin_act_kernel       sums the partials (float64, fixed order) + bias + sigmoid + Philox sample (rbm.py:198-213).
//   thin_pass_kernel<1>   one gibbs_hvh (rbm.py:242-248; GRBM :662-671): a workgroup holds whole rows of W, so its slice of
//                         v1 = act(h W^T + vbias) is COMPLETE inside the workgroup -- no partial, no second launch -- and
//                         the same rows, still in LDS, give its partial of v1 W at once: one read of W for propdown AND
//                         the following propup (only a [B, H] partial crosses workgroups).
//   thin_act_kernel       the hidden means / samples of that Gibbs step.
//   thin_update_kernel    rbm.py:392-419 + :347-365: S = [v0; nv]^T [ph; -nh] has rank 2B <= 64, so each row of S is formed
//                         in registers from 2B rank-1 terms and consumed on the spot by the update rule: W and W_speed are
//                         read and written once, S never exists in memory (single device), bias statistics and the bias
//                         half of the update ride along.
//
// W is read 2 + k times per CD-k step (propup, k Gibbs steps, update) and written once, against ~7 streams on the
// register-streaming GEMM path this replaces at B <= 32 (PMC, 19 937 -> 400 at batch 20: 245 MB per step,
// profiles/r05k_thin_pmc_traffic.json).  The products run on the bf16 matrix pipe at float32 accuracy
// (v_mfma_f32_32x32x16_bf16; the minibatch is one 32-row M tile): every float32 fragment -- read from the float32 LDS image of
// the workgroup's block of W, or straight from global memory in the positive phase -- is split EXACTLY into its three bf16
// pieces in registers on the way into the MFMA (mdbn_bf16x3.h: the six / three piece products and order of
// gemm_bf16x6_kernel); a first version on v_mfma_f32_32x32x2_f32 was bound by that pipe (1/16 of the bf16 rate).  The
// rank-2B statistics run on the VALU.  Same Philox addressing and activation arithmetic (act_quad) as every other path.
#include <hip/hip_runtime.h>
#include "mdbn_thin.h"
#include "mdbn_device.h"
#include "mdbn_bf16x3.h"

namespace mdbn {


extern __shared__ __align__(16) float th_smem[];

__device__ __forceinline__ float4 lds_read4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// index -> row of the training matrix (numpy-style negative values, clamped like every gather of the library)
__device__ __forceinline__ int64_t thin_src_row(const void* idx, int idx64, int64_t r, int64_t n_rows)
{
    if (!idx) return r;
    int64_t s = idx64 ? reinterpret_cast<const int64_t*>(idx)[r] : (int64_t)reinterpret_cast<const int32_t*>(idx)[r];
    if (s < 0) s += n_rows;
    return s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s);
}

// ------------------------------------------------------------------------------------------------------------------
// One pass over W (see the file comment).  The workgroup's rows of W, [r0, r1), nrows <= rpw <= 256; R16 / R32 = rpw
// rounded up to 16 / 32; K16 = ldh rounded up to 16.  NT2 = 32-column tiles of the upward product per wave.
// XP = pieces of the upward product's A operand: 3 (float32 visible values), 1 (0/1 visible samples: MODE 1, RBM).
//
// MODE 1: the whole block of W is staged ONCE into LDS as float32 (every load of the workgroup in flight together: the
//   stage runs at the memory system's rate, not at a round trip per sub-block) and both products read it from there --
//   phase 1 along the rows (k = hidden unit: two ds_read_b128 per fragment), phase 2 down the columns (k = visible
//   unit: eight ds_read_b32) -- splitting each fragment into its three bf16 pieces in registers on the way into the MFMA.
//   LDS (bytes): Wf [R16][PW] f32 | hKb [Bq][PH] bf16, later red [S1][Bq][R32] f32 over it | xP [XP][Bq][PXb] bf16.
// MODE 0: x is gathered and split into xP once; W streams from global memory straight into the fragments (eight
//   global_load_dword per lane and 16-row step, two steps in flight), no LDS image of W, no barrier in the loop.
// ------------------------------------------------------------------------------------------------------------------
#ifdef MDBN_STAMP   // diagnostic builds (scripts/experiments/thin_stamps.py): wall-clock stamps of every workgroup's phases
#define TH_STAMP(SLOT) do { if (MODE == 1 && a.stamps && threadIdx.x == 0) a.stamps[(int64_t)blockIdx.x * 16 + (SLOT)] = wall_clock64(); } while (0)
#else
#define TH_STAMP(SLOT) do {} while (0)
#endif

template <int MODE, int NT2, int XP>
__global__ __launch_bounds__(TH_NT) void thin_pass_kernel(ThinPassArgs a)
{
    TH_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Bq = a.Bq, PW = a.PW;
    const int q4 = (int)(a.ldh >> 2);
    const int g = blockIdx.x;
    const int r0 = g * a.rpw, r1 = min(a.V, r0 + a.rpw), nrows = r1 - r0;
    const int R16 = (a.rpw + 15) & ~15, R32 = (a.rpw + 31) & ~31;
    const int K16 = ((int)a.ldh + 15) & ~15;
    const int PH = K16 + 8, PXb = R16 + 8;
    const int ln = lane & 31, kh = lane >> 5;
    const int bA = min(ln, Bq - 1);         // rows >= Bq of the 32-row M tile read a duplicate (their results are never stored)

    float* Wf = th_smem;
    unsigned short* hKb = reinterpret_cast<unsigned short*>(th_smem + (MODE == 1 ? R16 * PW : 0));
    float* red = reinterpret_cast<float*>(hKb);
    const int ntile1 = R32 >> 5, S1 = 8 / ntile1 > 0 ? 8 / ntile1 : 1;
    const int hk_bytes = Bq * PH * 2, red_bytes = S1 * Bq * R32 * 4;
    unsigned short* xP = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(hKb) +
                                                           (MODE == 1 ? ((max(hk_bytes, red_bytes) + 15) & ~15) : 0));
    const int xp_plane = Bq * PXb;

    float cost = 0.f;
    if (MODE == 1) {
        // ---- stage: W rows -> Wf (pad rows / pad columns: zeros), chain state -> hKb (bf16: 0/1 samples are exact)
        // the visible epilogue's operands of this thread's first item, and the chain state, requested FIRST: their round trips
        // pass under the stage of W
        const int e_i = tid % R32, e_bq = tid / R32;
        const bool e_on = tid < R32 * (Bq >> 2) && e_i < nrows;
        float vb_pre = 0.f, tg_pre[4] = {0.f, 0.f, 0.f, 0.f};
        if (e_on) {
            vb_pre = a.vbias[r0 + e_i];
            if (a.last && a.target) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * e_bq + r < a.B) tg_pre[r] = a.target[(int64_t)(4 * e_bq + r) * a.ld_target + r0 + e_i];
            }
        }
        const int h4 = PH >> 2;                 // (PH = K16 + 8: a multiple of 4)
        constexpr int HKN = 9;                  // float4 of the chain state per thread: Bq * h4 <= 32 * 130 = 9 * 512 - ...
        float4 hv[HKN];
#pragma unroll
        for (int u = 0; u < HKN; ++u) {
            const int e = tid + TH_NT * u;
            const int b = e / h4, c4 = e - b * h4;
            hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Bq * h4 && b < a.B && c4 < q4) hv[u] = *reinterpret_cast<const float4*>(a.chain + (int64_t)b * a.ldh + 4 * c4);
        }
        // (wave w stages rows w, w + 8, ...; a lane the float4 columns lane, lane + 64: no division, unconditional loads
        //  at clamped addresses, eight rows = up to 16 loads per lane in flight)
        const int k4 = K16 >> 2;
        for (int rb = wave; rb < R16; rb += 64) {
            float4 v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = rb + 8 * u;
                const float* src = a.W + (int64_t)(r0 + min(row, nrows - 1)) * a.ldh;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int c4 = lane + 64 * c;
                    v[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < nrows && 64 * c < q4) {                       // (wave-uniform)
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
        TH_STAMP(1);
#pragma unroll
        for (int u = 0; u < HKN; ++u) {         // (bf16 image: the upper halves of the 0/1 floats)
            const int e = tid + TH_NT * u;
            const int b = e / h4, c4 = e - b * h4;
            if (e < Bq * h4) {
                uint2 w;
                w.x = (__builtin_bit_cast(unsigned, hv[u].x) >> 16) | (__builtin_bit_cast(unsigned, hv[u].y) & 0xffff0000u);
                w.y = (__builtin_bit_cast(unsigned, hv[u].z) >> 16) | (__builtin_bit_cast(unsigned, hv[u].w) & 0xffff0000u);
                *reinterpret_cast<uint2*>(hKb + b * PH + 4 * c4) = w;
            }
        }
        TH_STAMP(2);
        __syncthreads();
        TH_STAMP(3);

        // ---- phase 1: v1_pre[b][i] = sum_j h[b][j] W[r0 + i][j]: work item = (32-row tile of the block, K share)
        f32x16 acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
        const int t1 = wave % ntile1, ks = wave / ntile1;
        const bool has1 = ks < S1;
        if (has1) {
            const int nsteps = K16 >> 4, nper = (nsteps + S1 - 1) / S1;
            const int s_end = min(nsteps, (ks + 1) * nper);
            const int rowB = min(32 * t1 + ln, R16 - 1);
            for (int s = ks * nper; s < s_end; ++s) {
                const tu32x4 aw = *reinterpret_cast<const tu32x4*>(hKb + bA * PH + 16 * s + 8 * kh);
                const float4 w0 = lds_read4(Wf + rowB * PW + 16 * s + 8 * kh), w1 = lds_read4(Wf + rowB * PW + 16 * s + 8 * kh + 4);
                const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                tbf16x8 fb[3], fa[3];
                th_split8(f, fb);
                fa[0] = __builtin_bit_cast(tbf16x8, aw); fa[1] = fa[0]; fa[2] = fa[0];
                th_mma<1>(acc1, fa, fb);
            }
        }
        TH_STAMP(4);
        __syncthreads();                        // every wave is done reading hKb: red may overwrite it
        if (has1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (b < Bq) red[(ks * Bq + b) * R32 + 32 * t1 + ln] = acc1[r];
            }
        }
        __syncthreads();
        TH_STAMP(5);
        // ---- visible activation: thread = 4 rows x 1 column (one Philox block), rbm.py:226-240 / :650-658
        for (int e = tid; e < R32 * (Bq >> 2); e += TH_NT) {
            const int i = e % R32, bq = e / R32;
            const int col = r0 + i;
            const bool live = i < nrows;
            float vb_e = vb_pre, tg_e[4] = {tg_pre[0], tg_pre[1], tg_pre[2], tg_pre[3]};
            if (e != tid) {                     // (a second item per thread: only batches of more than 20 rows on wide blocks)
                vb_e = live ? a.vbias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tg_e[r] = (live && a.last && a.target && 4 * bq + r < a.B) ? a.target[(int64_t)(4 * bq + r) * a.ld_target + col] : 0.f;
            }
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sacc = 0.f;
                for (int w = 0; w < S1; ++w) sacc += red[(w * Bq + 4 * bq + r) * R32 + i];
                x[r] = sacc + vb_e;
            }
            uint32_t wa[4] = {0u, 0u, 0u, 0u};
            if (!a.gauss) philox_rows4(a.rng, a.rng.draw, a.rng.row_offset + (uint64_t)(4 * bq), (uint32_t)col, wa);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 4 * bq + r;
                const bool on = live && b < a.B;
                float m, sv;
                if (a.gauss) { m = x[r]; sv = m; }                         // error_free GRBM: the chain goes on from the mean (rbm.py:669)
                else { m = sigmoidf_(x[r]); sv = philox_u01(wa[r]) < m ? 1.0f : 0.0f; }
                if (on && a.last && a.target) {
                    if (a.gauss) { const float d = sigmoidf_(x[r]) - tg_e[r]; cost += d * d; }          // rbm.py:697
                    else cost += tg_e[r] * softplusf_(-x[r]) + (1.0f - tg_e[r]) * softplusf_(x[r]);      // rbm.py:479-480
                }
                if (on) {
                    a.nv[(int64_t)b * a.ldv + col] = m;
                    if (a.vs) a.vs[(int64_t)b * a.ldv + col] = sv;
                }
                if (i < R16) {
                    const float xv = on ? sv : 0.f;
                    unsigned short q1, q2, q3;
                    split3(xv, q1, q2, q3);
                    xP[b * PXb + i] = q1;
                    if (XP == 3) { xP[xp_plane + b * PXb + i] = q2; xP[2 * xp_plane + b * PXb + i] = q3; }
                }
            }
        }
        TH_STAMP(6);
        __syncthreads();
        TH_STAMP(7);
    } else {
        // ---- MODE 0: (the gather of x and the first steps of W are requested together: below)
    }

    // ---- phase 2: partial[b][j] = sum_i x[b][i] W[r0 + i][j]; 32-column tiles wave, wave + 8; 16 rows of W per step
    f32x16 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    const int nst2 = (nrows + 15) >> 4;
    int jt[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) jt[t] = min(32 * (wave + 8 * t) + ln, (int)a.ldh - 1);     // (lanes past ldh: a column nobody stores)
    if (32 * wave < (int)a.ldh) {
        if (MODE == 1) {
            for (int s = 0; s < nst2; ++s) {
                tbf16x8 fa[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fa[pl] = pl < XP ? *reinterpret_cast<const tbf16x8*>(xP + pl * xp_plane + bA * PXb + 16 * s + 8 * kh) : fa[0];
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    if (32 * (wave + 8 * t) < (int)a.ldh) {
                        const float* wp = Wf + (16 * s + 8 * kh) * PW + jt[t];
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = wp[e * PW];
                        tbf16x8 fb[3];
                        th_split8(f, fb);
                        th_mma<XP>(acc2[t], fa, fb);
                    }
                }
            }
        } else {
            // THREE steps of global loads in flight in three NAMED buffers, the loop unrolled by three (indexing one array
            // with the step's residue made hipcc wait for every single load).  Order of requests: the minibatch's row
            // indices, the first three steps of W, then -- the indices are back by now -- the rows of x themselves: the
            // gather's two dependent round trips pass beside W's first one instead of in front of it.
            float fA[NT2][8], fB[NT2][8], fC[NT2][8];
            auto issue = [&](int s, float (&f)[NT2][8]) {
#pragma unroll
                for (int t = 0; t < NT2; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = 16 * s + 8 * kh + e;
                        f[t][e] = a.W[(int64_t)(r0 + min(row, nrows - 1)) * a.ldh + jt[t]];     // (masked on use: consume)
                    }
            };
            auto consume = [&](int s, const float (&f)[NT2][8]) {
                tbf16x8 fa[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const tbf16x8*>(xP + pl * xp_plane + bA * PXb + 16 * s + 8 * kh);
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    if (32 * (wave + 8 * t) < (int)a.ldh) {
                        float fm[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) fm[e] = 16 * s + 8 * kh + e < nrows ? f[t][e] : 0.f;   // rows past the block: exact zeros
                        tbf16x8 fb[3];
                        th_split8(fm, fb);
                        th_mma<3>(acc2[t], fa, fb);
                    }
                }
            };
            constexpr int GN = 4;               // items of the [Bq][R16] tile of x per thread and round: 32 * 256 / 512 / 4 rounds at most
            const int items = Bq * R16;
            for (int e0 = 0; e0 < items; e0 += TH_NT * GN) {
                int64_t srow[GN];
#pragma unroll
                for (int u = 0; u < GN; ++u) {
                    const int e = e0 + tid + TH_NT * u;
                    const int b = min(e / R16, a.B - 1);
                    srow[u] = thin_src_row(a.idx, a.idx64, b, a.n_data);
                }
                if (e0 == 0) { issue(0, fA); if (nst2 > 1) issue(1, fB); if (nst2 > 2) issue(2, fC); }
                float xv[GN];
#pragma unroll
                for (int u = 0; u < GN; ++u) {
                    const int e = e0 + tid + TH_NT * u;
                    const int b = e / R16, i = e - b * R16;
                    xv[u] = (e < items && b < a.B && i < nrows) ? a.data[srow[u] * a.ld_data + r0 + i] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < GN; ++u) {
                    const int e = e0 + tid + TH_NT * u;
                    const int b = e / R16, i = e - b * R16;
                    if (e < items) {
                        if (b < a.B && i < nrows) a.v0_out[(int64_t)b * a.ldv + r0 + i] = xv[u];
                        unsigned short q1, q2, q3;
                        split3(xv[u], q1, q2, q3);
                        xP[b * PXb + i] = q1; xP[xp_plane + b * PXb + i] = q2; xP[2 * xp_plane + b * PXb + i] = q3;
                    }
                }
            }
            __syncthreads();
            for (int s = 0; s < nst2; s += 3) {
                consume(s, fA);
                if (s + 3 < nst2) issue(s + 3, fA);
                if (s + 1 < nst2) consume(s + 1, fB);
                if (s + 4 < nst2) issue(s + 4, fB);
                if (s + 2 < nst2) consume(s + 2, fC);
                if (s + 5 < nst2) issue(s + 5, fC);
            }
        }
    }
    else if (MODE == 0) {
        // (waves without a tile of the upward product still take part in the gather of x and in its barrier)
        const int items = Bq * R16;
        for (int e = tid; e < items; e += TH_NT) {
            const int b = e / R16, i = e - b * R16;
            float xv = 0.f;
            if (b < a.B && i < nrows) {
                xv = a.data[thin_src_row(a.idx, a.idx64, b, a.n_data) * a.ld_data + r0 + i];
                a.v0_out[(int64_t)b * a.ldv + r0 + i] = xv;
            }
            unsigned short q1, q2, q3;
            split3(xv, q1, q2, q3);
            xP[b * PXb + i] = q1; xP[xp_plane + b * PXb + i] = q2; xP[2 * xp_plane + b * PXb + i] = q3;
        }
        __syncthreads();
    }

    TH_STAMP(8);
    // the workgroup's partial of the upward product: rows 0 .. Bq - 1 (rows >= B are exact zeros)
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int j = 32 * (wave + 8 * t) + ln;
        if (j < (int)a.ldh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (b < Bq) a.part[((int64_t)g * Bq + b) * a.ldh + j] = acc2[t][r];
            }
        }
    }
    if (MODE == 0 && g == 0) {              // pad columns of the gathered rows stay zero
        for (int64_t e = tid; e < (int64_t)a.B * (a.ldv - a.V); e += TH_NT) {
            const int64_t b = e / (a.ldv - a.V), c = e - b * (a.ldv - a.V);
            a.v0_out[b * a.ldv + a.V + c] = 0.f;
        }
    }
    TH_STAMP(9);
    if (MODE == 1 && a.cost_partials) {
        __syncthreads();
        const float tot = block_sum(cost, red);
        if (tid == 0) a.cost_partials[g] = tot;
    }
    TH_STAMP(10);
}

template <int MODE, int NT2, int XP>
static hipError_t launch_thin_pass_t(const ThinPassArgs& a, int lds, hipStream_t s)
{
    auto kern = thin_pass_kernel<MODE, NT2, XP>;
    static bool attr_done = false;          // (per instantiation)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TH_MAX_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.G), dim3(TH_NT), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_thin_pass(int mode, const ThinPassArgs& a, const ThinGeom& t, hipStream_t s)
{
    if (mode == 0) {
        if (t.nt2 == 1) return launch_thin_pass_t<0, 1, 3>(a, t.lds_up, s);
        if (t.nt2 == 2) return launch_thin_pass_t<0, 2, 3>(a, t.lds_up, s);
        return hipErrorInvalidValue;
    }
    if (a.gauss) {
        if (t.nt2 == 1) return launch_thin_pass_t<1, 1, 3>(a, t.lds_pass, s);
        if (t.nt2 == 2) return launch_thin_pass_t<1, 2, 3>(a, t.lds_pass, s);
    } else {
        if (t.nt2 == 1) return launch_thin_pass_t<1, 1, 1>(a, t.lds_pass, s);
        if (t.nt2 == 2) return launch_thin_pass_t<1, 2, 1>(a, t.lds_pass, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------------
// Sum of the G partials + bias + activation + sampling.  Workgroup = (4-row group, LW columns); its 1024 / LW lane groups
// take the partials p = group, group + 1024 / LW, ... in batches of eight partials x four rows of loads, sum in float64 and
// round ONCE, as act_epilogue_kernel does with split-K slabs; group 0 then adds the group sums in order (deterministic).
// LW = 32 (half-waves) for many partials: at G = 256 every lane has ONE batch of 32 loads -- one round trip to the partials
// for the whole launch (19 937 -> 400: 6.5 -> 5.6 us); LW = 64 for few (784 -> 500, G = 49: 5.0 us against 5.8 with LW = 32).
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, int LW>       // MODE: act_quad's bit 0 = a sample is wanted (hidden units are Bernoulli: bit 1, gauss, is never set here)
__global__ __launch_bounds__(TH_ACT_NT) void thin_act_kernel(ThinActArgs a)
{
    constexpr int NG = TH_ACT_NT / LW;
    __shared__ double redd[NG][4][LW];
    const int l32 = threadIdx.x % LW, hw = threadIdx.x / LW;
    const EpiArgs& e = a.e;
    const int ncc = (int)((e.ld + LW - 1) / LW);
    const int rg = blockIdx.x / ncc, cc = blockIdx.x - rg * ncc;
    const int col = LW * cc + l32, r0 = 4 * rg;
    const int colc = min(col, (int)e.ld - 1);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t pstride = (int64_t)a.Bq * e.ld;
    const float* base = a.part + (int64_t)r0 * e.ld + colc;
    for (int p0 = hw; p0 < a.G; p0 += NG * 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = min(p0 + NG * u, a.G - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[u][r] = base[(int64_t)p * pstride + (int64_t)r * e.ld];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += p0 + NG * u < a.G ? (double)v[u][r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) redd[hw][r][l32] = acc[r];
    __syncthreads();
    if (hw == 0 && col < (int)e.ld) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4                      // (all reads at once would not fit the 128 registers of a 16-wave workgroup)
        for (int w = 0; w < NG; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += redd[w][r][l32];
        // hidden units: sigmoid + Bernoulli (rbm.py:198-213); act_quad's arithmetic and Philox words, without its cost /
        // plane / column-sum cases (with those compiled in, the kernel spilled at the 128 registers 16 waves leave a thread)
        const bool live = col < e.cols;
        const float bias = live ? e.bias[col] : 0.f;
        uint32_t wa[4] = {0u, 0u, 0u, 0u};
        if (MODE & 1) philox_rows4(e.rng, e.rng.draw, e.rng.row_offset + (uint64_t)r0, (uint32_t)col, wa);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r0 + r >= e.rows) break;
            const int64_t off = (int64_t)(r0 + r) * e.ld + col;
            float m = sigmoidf_((float)t[r] + bias);
            float sv = (MODE & 1) ? (philox_u01(wa[r]) < m ? 1.0f : 0.0f) : 0.f;
            if (!live) { m = 0.f; sv = 0.f; }                      // pad columns stay zero
            if (e.mean) e.mean[off] = m * e.mean_scale;
            if ((MODE & 1) && e.sample) e.sample[off] = sv;
        }
    }
}

hipError_t launch_thin_act(const ThinActArgs& a, hipStream_t s)
{
    if (a.e.gauss) return hipErrorInvalidValue;
    const bool sample = a.e.sample != nullptr || a.e.sample_plane != nullptr;
    const int lw = a.G >= 128 ? 32 : 64;
    const int ncc = (int)((a.e.ld + lw - 1) / lw);
    const dim3 grid((a.Bq >> 2) * ncc), block(TH_ACT_NT);
    if (lw == 32) {
        if (sample) hipLaunchKernelGGL((thin_act_kernel<1, 32>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((thin_act_kernel<0, 32>), grid, block, 0, s, a);
    } else {
        if (sample) hipLaunchKernelGGL((thin_act_kernel<1, 64>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((thin_act_kernel<0, 64>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Statistics + update (see the file comment).  A wave owns 64 * CW columns (chunk w % ncw) and walks the rows
// w / ncw, + 8 / ncw, ... of the workgroup's range; a lane keeps the 2 Bq x CW values of [ph; -nh] of its columns in
// registers, the row's 2 Bq values of [v0; nv] come from LDS as wave-uniform (broadcast) reads.
// ------------------------------------------------------------------------------------------------------------------
template <int CW> struct ThinVec;
template <> struct ThinVec<4> {
    typedef float4 T;
    static __device__ __forceinline__ void get(const T& v, float (&x)[4]) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
    static __device__ __forceinline__ T make(const float (&x)[4]) { return make_float4(x[0], x[1], x[2], x[3]); }
};
template <> struct ThinVec<2> {
    typedef float2 T;
    static __device__ __forceinline__ void get(const T& v, float (&x)[2]) { x[0] = v.x; x[1] = v.y; }
    static __device__ __forceinline__ T make(const float (&x)[2]) { return make_float2(x[0], x[1]); }
};

#ifndef TH_UPD_NT
#define TH_UPD_NT 0         // 1: W / W_speed streamed with non-temporal loads and stores (A/B: profiles/r05i_thin_update_variants.log)
#endif
template <typename VT>
__device__ __forceinline__ VT th_stream_load(const float* p)
{
#if TH_UPD_NT
    typedef float nvec __attribute__((ext_vector_type(sizeof(VT) / 4)));
    const nvec t = __builtin_nontemporal_load(reinterpret_cast<const nvec*>(p));
    return __builtin_bit_cast(VT, t);
#else
    return *reinterpret_cast<const VT*>(p);
#endif
}
template <typename VT>
__device__ __forceinline__ void th_stream_store(float* p, const VT& v)
{
#if TH_UPD_NT
    typedef float nvec __attribute__((ext_vector_type(sizeof(VT) / 4)));
    __builtin_nontemporal_store(__builtin_bit_cast(nvec, v), reinterpret_cast<nvec*>(p));
#else
    *reinterpret_cast<VT*>(p) = v;
#endif
}

template <int NB, int CW, bool WC>
__global__ __launch_bounds__(TH_UNT) void thin_update_kernel(ThinUpdArgs a)
{
    constexpr int Bq = 4 * NB, R2 = 2 * Bq;
    constexpr int NW = TH_UNT / 64;
#ifndef TH_UPD_PD
#define TH_UPD_PD 8
#endif
    constexpr int PD = CW == 4 ? 4 : TH_UPD_PD;                // rows whose W / W_speed loads are in flight together per wave
    typedef typename ThinVec<CW>::T VT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int r0 = g * a.rpw, r1 = min(a.V, r0 + a.rpw), nrows = r1 - r0;
    float* x2T = th_smem;                   // [rpw][R2]: row i = the 2 Bq values [v0; nv][., r0 + i] (pad rows of the batch: 0)

    int ncw = 1;
    while (ncw * 64 * CW < (int)a.ldh) ncw *= 2;              // 1 | 2 | 4 column chunks (thin_geom: ldh <= 512)
    const int cc = wave % ncw, rl = wave / ncw, nrl = NW / ncw > 0 ? NW / ncw : 1;
    const int npass = ncw > NW ? ncw / NW : 1;                // (CW = 2, ldh > 512 cannot happen; kept general)
    const UpdEpi& u = a.upd;

    // this lane's columns of [ph; -nh], requested first (L2-resident: the activation kernels just wrote them)
    const int j = 64 * CW * cc + CW * lane;
    const bool jok = j < (int)a.ldh && rl < nrl;
    const int jc = j < (int)a.ldh ? j : 0;
    const float two_lr_l1 = upd_two_lr_l1(u.lr, u.l1);
    const float decay = upd_decay(u.lr, u.l2);
    const float* w0base = u.W0 ? u.W0 : u.W;

    float p2[R2][CW];
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int half = r >= Bq, b = r - half * Bq;
        float t[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) t[c] = 0.f;
        if (b < a.B && jok) ThinVec<CW>::get(*reinterpret_cast<const VT*>(a.P2 + (int64_t)(half * a.B + b) * a.ldh + jc), t);
#pragma unroll
        for (int c = 0; c < CW; ++c) p2[r][c] = t[c];
    }
    for (int e = tid; e < R2 * a.rpw; e += TH_UNT) {
        const int r = e / a.rpw, i = e - r * a.rpw;           // (consecutive threads: consecutive columns of one row of V2)
        const int half = r >= Bq, b = r - half * Bq;
        float v = 0.f;
        if (b < a.B && i < nrows) v = a.V2[(int64_t)(half * a.B + b) * a.ldv + r0 + i];
        x2T[i * R2 + r] = v;
    }
    __syncthreads();
    (void)npass;

    if (jok) {
        for (int i = rl; i < nrows; i += PD * nrl) {
            // (requesting the first rows before the prologue above was measured: no gain, profiles/r05n)
            VT w[PD], sp[PD], w0[PD];
            if (a.do_upd) {
#pragma unroll
                for (int q = 0; q < PD; ++q) {
                    const int ic = min(i + q * nrl, nrows - 1);
                    const int64_t off = (int64_t)(r0 + ic) * a.ldh + j;
                    w[q] = th_stream_load<VT>(u.W + off);
                    sp[q] = th_stream_load<VT>(u.Ws + off);
                    if (WC) w0[q] = *reinterpret_cast<const VT*>(w0base + off);
                }
            }
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ii = i + q * nrl;
                if (ii >= nrows) break;
                float st[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) st[c] = 0.f;
                const float* xr = x2T + ii * R2;
#pragma unroll
                for (int r4 = 0; r4 < R2 / 4; ++r4) {
                    const float4 xv = lds_read4(xr + 4 * r4);
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        st[c] = fmaf(xv.x, p2[4 * r4 + 0][c], st[c]);
                        st[c] = fmaf(xv.y, p2[4 * r4 + 1][c], st[c]);
                        st[c] = fmaf(xv.z, p2[4 * r4 + 2][c], st[c]);
                        st[c] = fmaf(xv.w, p2[4 * r4 + 3][c], st[c]);
                    }
                }
                const int64_t off = (int64_t)(r0 + ii) * a.ldh + j;
                if (a.do_upd) {
                    float wv[CW], sv[CW], w0v[CW], wn[CW], sn[CW];
                    ThinVec<CW>::get(w[q], wv); ThinVec<CW>::get(sp[q], sv);
                    if (WC) ThinVec<CW>::get(w0[q], w0v);
#pragma unroll
                    for (int c = 0; c < CW; ++c) {            // update_rule4's arithmetic, element by element (same helpers)
                        float gr = upd_grad(st[c], u.inv_bs, WC ? u.wc : 0.f, WC ? w0v[c] : 0.f);
                        float m = decay;
                        if (u.l1 != 0.0f) {
                            const float shrink = upd_shrink(two_lr_l1, wv[c]);
                            gr = __fdiv_rn(gr, shrink);
                            m = __fdiv_rn(decay, shrink);
                        }
                        sn[c] = upd_speed(gr, sv[c], u.mu);
                        wn[c] = upd_param(wv[c], m, sv[c], u.lr);
                    }
                    th_stream_store<VT>(u.W + off, ThinVec<CW>::make(wn));
                    th_stream_store<VT>(u.Ws + off, ThinVec<CW>::make(sn));
                    if (u.Wp) {
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            unsigned short q1, q2, q3;
                            split3(wn[c], q1, q2, q3);
                            u.Wp[off + c] = q1; u.Wp[u.wp_stride + off + c] = q2; u.Wp[2 * u.wp_stride + off + c] = q3;
                        }
                    }
                } else {
                    *reinterpret_cast<VT*>(a.S + off) = ThinVec<CW>::make(st);
                }
            }
        }
    }

    // s_v (rbm.py:417) of this workgroup's visible units, and their bias update
    for (int i = tid; i < nrows; i += TH_UNT) {
        float t = 0.f;
        for (int b = 0; b < a.B; ++b) t += x2T[i * R2 + b] - x2T[i * R2 + Bq + b];
        const int64_t col = r0 + i;
        a.s_v[col] = t;
        if (a.do_upd) {
            const BiasUpd& bu = a.bu;
            const float sp = bu.vbs[col], p0 = bu.vb[col];
            bu.vbs[col] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
            bu.vb[col] = upd_param(p0, 1.0f, sp, bu.lr);
        }
    }
    if (g == 0) {
        // s_h (rbm.py:416), the hidden bias update, the pad entries of s_h / s_v, the cost total
        // (from the registers: the waves of row lane 0 hold [ph; -nh] of their columns; summed rows 0 .. B - 1 of ph, then of -nh)
        if (jok && rl == 0) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < R2; ++r) t += p2[r][c];
                const int64_t jj = j + c;
                if (jj >= a.H) t = 0.f;
                a.s_h[jj] = t;
                if (a.do_upd && jj < a.H) {
                    const BiasUpd& bu = a.bu;
                    const float sp = bu.hbs[jj], p0 = bu.hb[jj];
                    bu.hbs[jj] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
                    bu.hb[jj] = upd_param(p0, 1.0f, sp, bu.lr);
                }
            }
        }
        for (int64_t c = a.V + tid; c < a.ldv; c += TH_UNT) a.s_v[c] = 0.f;
        if (wave == 0) {
            float t = 0.f;
            for (int k0 = lane; k0 < a.n_cost; k0 += 256) {        // (four loads in flight; lane order = the order of the sums)
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = k0 + 64 * q < a.n_cost ? a.cost_partials[k0 + 64 * q] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) t += v[q];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) {
                a.cost[0] = t; a.cost[1] = 0.f; a.cost[2] = 0.f; a.cost[3] = 0.f;
                if (a.do_upd && a.bu.cost_out) a.bu.cost_out[0] = t * a.bu.cost_scale;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// update(t) + positive phase(t + 1) in ONE pass over W (the trainers announce the next minibatch:
// mdbn_cd_args.next_indexes).  thin_update_kernel's row loop on the PASS geometry (G workgroups of 512 threads, rpw rows):
// every updated row W' goes to global memory and into a float32 LDS image of the block; when the block is done the
// workgroup multiplies x' = data[next_indexes][:, block] into it exactly as thin_pass_kernel<1>'s upward product does and
// writes ONE partial of x' W' -- the next step starts at its first activation kernel.  v0' overwrites rows 0..B-1 of V2
// (this workgroup's columns only, after it has staged its block of [v0; nv]).
// LDS (bytes): Wf [R16][PW] f32 | x2T [rpw][2 Bq] f32 | xP [3][Bq][R16 + 8] bf16.
// ------------------------------------------------------------------------------------------------------------------
template <int NB, bool WC, int NT2>
__global__ __launch_bounds__(TH_NT) void thin_update_ahead_kernel(ThinUpdArgs a)
{
    constexpr int CW = 2, Bq = 4 * NB, R2 = 2 * Bq, NW = TH_NT / 64, PD = NB >= 7 ? 4 : 8;      // (rows in flight: 256 VGPRs)
    typedef typename ThinVec<CW>::T VT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, kh = lane >> 5;
    const int g = blockIdx.x;
    const int r0 = g * a.rpw, r1 = min(a.V, r0 + a.rpw), nrows = r1 - r0;
    const int R16 = (a.rpw + 15) & ~15, PXb = R16 + 8, PW = a.PW;
    const int K16 = ((int)a.ldh + 15) & ~15;
    float* Wf = th_smem;
    float* x2T = Wf + R16 * PW;
    unsigned short* xP = reinterpret_cast<unsigned short*>(x2T + a.rpw * R2);
    const int xp_plane = Bq * PXb;

    int ncw = 1;
    while (ncw * 64 * CW < (int)a.ldh) ncw *= 2;              // 1 | 2 | 4 column chunks of 128
    const int cc = wave % ncw, rl = wave / ncw, nrl = NW / ncw;
    const UpdEpi& u = a.upd;
    const int j = 64 * CW * cc + CW * lane;
    const bool jok = j < (int)a.ldh;
    const int jc = jok ? j : 0;

    constexpr int GN = 4;                   // items of the [Bq][R16] tile of x' per thread and round
    const int items = Bq * R16;
    int64_t srow[GN];
#pragma unroll
    for (int q = 0; q < GN; ++q) srow[q] = thin_src_row(a.next_idx, a.idx64, min((tid + TH_NT * q) / R16, a.B - 1), a.n_data);

    float p2[R2][CW];
#pragma unroll
    for (int r = 0; r < R2; ++r) {
        const int half = r >= Bq, b = r - half * Bq;
        float t[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) t[c] = 0.f;
        if (b < a.B && jok) ThinVec<CW>::get(*reinterpret_cast<const VT*>(a.P2 + (int64_t)(half * a.B + b) * a.ldh + jc), t);
#pragma unroll
        for (int c = 0; c < CW; ++c) p2[r][c] = t[c];
    }
    for (int e = tid; e < R2 * a.rpw; e += TH_NT) {
        const int r = e / a.rpw, i = e - r * a.rpw;
        const int half = r >= Bq, b = r - half * Bq;
        float v = 0.f;
        if (b < a.B && i < nrows) v = a.V2[(int64_t)(half * a.B + b) * a.ldv + r0 + i];
        x2T[i * R2 + r] = v;
    }
    // pad rows / pad columns of the image: zeros (the row loop writes columns < ldh of rows < nrows)
    for (int e = tid; e < R16 * (K16 >> 2); e += TH_NT) {
        const int row = e / (K16 >> 2), c4 = e - row * (K16 >> 2);
        if (row >= nrows || 4 * c4 >= (int)a.ldh) *reinterpret_cast<float4*>(Wf + row * PW + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                        // this workgroup's block of [v0; nv] is in LDS: rows 0..B-1 of V2 may be overwritten
    // x' = data[next_indexes][:, block]: the loads go out here and land while the first rows of W stream in (consumed after
    // the row loop); the row indices were requested at the top of the kernel
    float xv[GN];
#pragma unroll
    for (int q = 0; q < GN; ++q) {
        const int e = tid + TH_NT * q;
        const int b = e / R16, i = e - b * R16;
        xv[q] = (e < items && b < a.B && i < nrows) ? a.data[srow[q] * a.ld_data + r0 + i] : 0.f;
    }

    const float two_lr_l1 = upd_two_lr_l1(u.lr, u.l1);
    const float decay = upd_decay(u.lr, u.l2);
    const float* w0base = u.W0 ? u.W0 : u.W;
    if (jok) {
        for (int i = rl; i < nrows; i += PD * nrl) {
            VT w[PD], sp[PD], w0[PD];
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ic = min(i + q * nrl, nrows - 1);
                const int64_t off = (int64_t)(r0 + ic) * a.ldh + j;
                w[q] = th_stream_load<VT>(u.W + off);
                sp[q] = th_stream_load<VT>(u.Ws + off);
                if (WC) w0[q] = *reinterpret_cast<const VT*>(w0base + off);
            }
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int ii = i + q * nrl;
                if (ii >= nrows) break;
                float st[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) st[c] = 0.f;
                const float* xr = x2T + ii * R2;
#pragma unroll
                for (int r4 = 0; r4 < R2 / 4; ++r4) {
                    const float4 xv = lds_read4(xr + 4 * r4);
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        st[c] = fmaf(xv.x, p2[4 * r4 + 0][c], st[c]);
                        st[c] = fmaf(xv.y, p2[4 * r4 + 1][c], st[c]);
                        st[c] = fmaf(xv.z, p2[4 * r4 + 2][c], st[c]);
                        st[c] = fmaf(xv.w, p2[4 * r4 + 3][c], st[c]);
                    }
                }
                const int64_t off = (int64_t)(r0 + ii) * a.ldh + j;
                float wv[CW], sv[CW], w0v[CW], wn[CW], sn[CW];
                ThinVec<CW>::get(w[q], wv); ThinVec<CW>::get(sp[q], sv);
                if (WC) ThinVec<CW>::get(w0[q], w0v);
#pragma unroll
                for (int c = 0; c < CW; ++c) {                // update_rule4's arithmetic, element by element (same helpers)
                    float gr = upd_grad(st[c], u.inv_bs, WC ? u.wc : 0.f, WC ? w0v[c] : 0.f);
                    float mm = decay;
                    if (u.l1 != 0.0f) {
                        const float shrink = upd_shrink(two_lr_l1, wv[c]);
                        gr = __fdiv_rn(gr, shrink);
                        mm = __fdiv_rn(decay, shrink);
                    }
                    sn[c] = upd_speed(gr, sv[c], u.mu);
                    wn[c] = upd_param(wv[c], mm, sv[c], u.lr);
                }
                th_stream_store<VT>(u.W + off, ThinVec<CW>::make(wn));
                th_stream_store<VT>(u.Ws + off, ThinVec<CW>::make(sn));
                *reinterpret_cast<VT*>(Wf + ii * PW + j) = ThinVec<CW>::make(wn);          // the NEW row, for the next positive phase
                if (u.Wp) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        unsigned short q1, q2, q3;
                        split3(wn[c], q1, q2, q3);
                        u.Wp[off + c] = q1; u.Wp[u.wp_stride + off + c] = q2; u.Wp[2 * u.wp_stride + off + c] = q3;
                    }
                }
            }
        }
    }

    auto put_x = [&](int e, float v) {
        const int b = e / R16, i = e - b * R16;
        if (b < a.B && i < nrows) a.v0_next[(int64_t)b * a.ldv + r0 + i] = v;
        unsigned short q1, q2, q3;
        split3(v, q1, q2, q3);
        xP[b * PXb + i] = q1; xP[xp_plane + b * PXb + i] = q2; xP[2 * xp_plane + b * PXb + i] = q3;
    };
#pragma unroll
    for (int q = 0; q < GN; ++q)
        if (tid + TH_NT * q < items) put_x(tid + TH_NT * q, xv[q]);
    for (int e0 = TH_NT * GN; e0 < items; e0 += TH_NT * GN) {           // (tiles of more than 2048 items: further rounds)
        int64_t sr[GN];
#pragma unroll
        for (int q = 0; q < GN; ++q) sr[q] = thin_src_row(a.next_idx, a.idx64, min((e0 + tid + TH_NT * q) / R16, a.B - 1), a.n_data);
        float xw[GN];
#pragma unroll
        for (int q = 0; q < GN; ++q) {
            const int e = e0 + tid + TH_NT * q;
            const int b = e / R16, i = e - b * R16;
            xw[q] = (e < items && b < a.B && i < nrows) ? a.data[sr[q] * a.ld_data + r0 + i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < GN; ++q)
            if (e0 + tid + TH_NT * q < items) put_x(e0 + tid + TH_NT * q, xw[q]);
    }

    // s_v (rbm.py:417) of this workgroup's visible units and their bias update; workgroup 0: s_h, hidden biases, cost
    for (int i = tid; i < nrows; i += TH_NT) {
        float t = 0.f;
        for (int b = 0; b < a.B; ++b) t += x2T[i * R2 + b] - x2T[i * R2 + Bq + b];
        const int64_t col = r0 + i;
        a.s_v[col] = t;
        const BiasUpd& bu = a.bu;
        const float sp = bu.vbs[col], p0 = bu.vb[col];
        bu.vbs[col] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
        bu.vb[col] = upd_param(p0, 1.0f, sp, bu.lr);
    }
    if (g == 0) {
        if (jok && rl == 0) {
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < R2; ++r) t += p2[r][c];
                const int64_t jj = j + c;
                if (jj >= a.H) t = 0.f;
                a.s_h[jj] = t;
                if (jj < a.H) {
                    const BiasUpd& bu = a.bu;
                    const float sp = bu.hbs[jj], p0 = bu.hb[jj];
                    bu.hbs[jj] = upd_speed(upd_scale(t, bu.inv_rows), sp, bu.mu);
                    bu.hb[jj] = upd_param(p0, 1.0f, sp, bu.lr);
                }
            }
        }
        for (int64_t c = a.V + tid; c < a.ldv; c += TH_NT) a.s_v[c] = 0.f;
        if (wave == NW - 1) {
            float t = 0.f;
            for (int k0 = lane; k0 < a.n_cost; k0 += 256) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = k0 + 64 * q < a.n_cost ? a.cost_partials[k0 + 64 * q] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) t += v[q];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0) {
                a.cost[0] = t; a.cost[1] = 0.f; a.cost[2] = 0.f; a.cost[3] = 0.f;
                if (a.bu.cost_out) a.bu.cost_out[0] = t * a.bu.cost_scale;
            }
        }
    }
    __syncthreads();                        // the image of the updated block and the planes of x' are complete

    // ---- the next step's positive phase over this block: partial[b][j] = sum_i x'[b][i] W'[r0 + i][j]
    f32x16 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    const int bA = min(ln, Bq - 1);
    if (32 * wave < (int)a.ldh) {
        const int nst2 = (nrows + 15) >> 4;
        for (int s = 0; s < nst2; ++s) {
            tbf16x8 fa[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const tbf16x8*>(xP + pl * xp_plane + bA * PXb + 16 * s + 8 * kh);
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                if (32 * (wave + 8 * t) < (int)a.ldh) {
                    const int jt = min(32 * (wave + 8 * t) + ln, (int)a.ldh - 1);
                    const float* wp = Wf + (16 * s + 8 * kh) * PW + jt;
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = wp[e * PW];
                    tbf16x8 fb[3];
                    th_split8(f, fb);
                    th_mma<3>(acc2[t], fa, fb);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int jj = 32 * (wave + 8 * t) + ln;
        if (jj < (int)a.ldh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (b < Bq) a.part_next[((int64_t)g * Bq + b) * a.ldh + jj] = acc2[t][r];
            }
        }
    }
}

template <int NB, int NT2>
static hipError_t launch_thin_update_ahead_t(const ThinUpdArgs& a, int lds, hipStream_t s)
{
    if (a.upd.wc != 0.f) {
        auto kern = thin_update_ahead_kernel<NB, true, NT2>;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TH_MAX_LDS);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3(a.G), dim3(TH_NT), lds, s, a);
    } else {
        auto kern = thin_update_ahead_kernel<NB, false, NT2>;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, TH_MAX_LDS);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(kern, dim3(a.G), dim3(TH_NT), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_thin_update_ahead(const ThinUpdArgs& a, const ThinGeom& t, hipStream_t s)
{
    if (!a.do_upd || t.lds_ahead <= 0) return hipErrorInvalidValue;
#define TH_AHEAD_CASE(NBV) case NBV: return t.nt2 == 1 ? launch_thin_update_ahead_t<NBV, 1>(a, t.lds_ahead, s) : launch_thin_update_ahead_t<NBV, 2>(a, t.lds_ahead, s)
    switch (t.Bq >> 2) {
    TH_AHEAD_CASE(1); TH_AHEAD_CASE(2); TH_AHEAD_CASE(3); TH_AHEAD_CASE(4);
    TH_AHEAD_CASE(5); TH_AHEAD_CASE(6); TH_AHEAD_CASE(7); TH_AHEAD_CASE(8);
    }
#undef TH_AHEAD_CASE
    return hipErrorInvalidValue;
}

template <int NB, int CW>
static hipError_t launch_thin_update_t(const ThinUpdArgs& a, int lds, hipStream_t s)
{
    // the weight-cost term (and its W0 stream) only where it is on
    if (a.do_upd && a.upd.wc != 0.f) hipLaunchKernelGGL((thin_update_kernel<NB, CW, true>), dim3(a.G), dim3(TH_UNT), lds, s, a);
    else hipLaunchKernelGGL((thin_update_kernel<NB, CW, false>), dim3(a.G), dim3(TH_UNT), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_thin_update(const ThinUpdArgs& a, const ThinGeom& t, hipStream_t s)
{
    switch (t.Bq >> 2) {
    case 1: return launch_thin_update_t<1, 4>(a, t.lds_upd, s);
    case 2: return launch_thin_update_t<2, 4>(a, t.lds_upd, s);
    case 3: return launch_thin_update_t<3, 4>(a, t.lds_upd, s);
    case 4: return launch_thin_update_t<4, 2>(a, t.lds_upd, s);
    case 5: return launch_thin_update_t<5, 2>(a, t.lds_upd, s);
    case 6: return launch_thin_update_t<6, 2>(a, t.lds_upd, s);
    case 7: return launch_thin_update_t<7, 2>(a, t.lds_upd, s);
    case 8: return launch_thin_update_t<8, 2>(a, t.lds_upd, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mdbn
