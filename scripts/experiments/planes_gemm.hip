// Experiment (not product): GEMM on pre-split bf16 planes.  Operands live in HBM as three bf16 planes
// per f32 matrix (x = p1 + p2 + p3 exactly); the kernel only COPIES them into LDS (LDS-DMA,
// global_load_lds_dwordx4) and issues the six piece products on v_mfma_f32_32x32x16_bf16.
//   ROW operand: planes [rows][ld], k contiguous  -> LDS image [128 rows][32 k], read by ds_read_b128
//   COL operand: planes [k][ld], rows contiguous  -> LDS image [32 k][128 rows], read by ds_read_b64_tr_b16
// 128x128 tile, 32-deep stages, 3-stage LDS ring (144 KB), 4 MFMA waves + LW loader waves.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef LORDER
#define LORDER 0    // loaders: 0 = issue stage it+2, wait, barrier; 1 = wait, barrier, issue stage it+3 (issue off the barrier's path)
#endif
#ifndef LPRIO
#define LPRIO 0     // s_setprio of the loader waves
#endif
#ifndef AUXBITS
#define AUXBITS 0   // cache-policy bits of the LDS-DMA loads (1 sc0, 2 nt, 16 sc1)
#endif
#ifndef SCHED
#define SCHED 0 // 0: compiler's order; 1: next step's fragment reads interleaved under this step's MFMAs; 2: all reads in front
#endif
#ifndef ABL
#define ABL 0   // diagnostic: 1 no DMA issued, 2 no MFMA (fragment reads kept alive), 3 no fragment reads, 4 consumers only pass barriers
#endif
enum { ROW = 0, COL = 1 };
constexpr int PLANE = 8192, STAGE = 6 * PLANE, NSTAGE = 3;

struct PArgs {
    const unsigned short* A; int64_t lda, pa;       // plane stride in elements
    const unsigned short* B; int64_t ldb, pb;
    float* C; int64_t ldc, slab_stride;
    int M, N, K, kchunk, splitk, tiles_m, tiles_n;
    unsigned long long* dbg;      // [grid][4]: s_memtime at entry / loop start / loop end, s_memrealtime span
};

__device__ __forceinline__ void glds16(const void* g, unsigned lds_off, char* smem)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, AUXBITS);
}

// One loader wave's share of the staging: instructions q = w, w + LW, ... of the NQ = 8 * (AP + 3) per stage.
// MS = MFMA shape the LDS images are swizzled for: 32 (v_mfma_f32_32x32x16_bf16) or 16 (v_mfma_f32_16x16x32_bf16)
__device__ __forceinline__ int row_swz(int row, int ms) { return ms == 32 ? ((row >> 2) & 3) : ((4 - ((row >> 2) & 3)) & 3); }
__device__ __forceinline__ int col_swz(int k, int ms) { return ms == 32 ? ((k & 3) << 2) : (((k & 3) | (((k >> 3) & 1) << 2)) << 1); }

template <int LA, int LB, int AP, int LW, int MS = 32>
__device__ __forceinline__ void loader(const PArgs& g, char* smem, int w, int lane, int m0, int n0, int kbeg, int nt)
{
    constexpr int NQ = 8 * (AP + 3), PER = NQ / LW;
    const char* src[PER];
    unsigned dst[PER];
    int64_t step[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int q = w + j * LW;
        const bool isA = q < 8 * AP;
        const int plane = isA ? q / 8 : (q - 8 * AP) / 8, sub = q & 7;
        const int lay = isA ? LA : LB;
        const unsigned short* base = isA ? g.A + plane * g.pa : g.B + plane * g.pb;
        const int64_t ld = isA ? g.lda : g.ldb;
        const int mn0 = isA ? m0 : n0;
        dst[j] = (isA ? plane : 3 + plane) * PLANE + sub * 1024;
        if (lay == ROW) {       // 16 rows x 64 B per instruction; phys chunk = c ^ ((row >> 2) & 3)
            const int row = 16 * sub + (lane >> 2);
            const int c = (lane & 3) ^ row_swz(row, MS);
            src[j] = reinterpret_cast<const char*>(base + (int64_t)(mn0 + row) * ld + kbeg + 8 * c);
            step[j] = 64;
        } else {                // 4 k-rows x 256 B per instruction; phys chunk = ch ^ ((k & 3) << 2)
            const int k = 4 * sub + (lane >> 4);
            const int ch = (lane & 15) ^ col_swz(k, MS);
            src[j] = reinterpret_cast<const char*>(base + (int64_t)(kbeg + k) * ld + mn0 + 8 * ch);
            step[j] = 64 * ld;
        }
    }
#define ISSUE(T)                                                                              \
    do {                                                                                      \
        const unsigned so = ((T) % NSTAGE) * STAGE;                                           \
        _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                     \
            if (ABL != 1) glds16(src[j], so + dst[j], smem);                                  \
            src[j] += step[j];                                                                \
        }                                                                                     \
    } while (0)
    if (LPRIO) __builtin_amdgcn_s_setprio(LPRIO);
#if LORDER == 0
    ISSUE(0);
    if (nt > 1) { ISSUE(1); }
    if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nt; ++it) {
        if (it + 2 < nt) {
            ISSUE(it + 2);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");     // stage it + 1 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
#else
    // three stages in flight; the refill of a slot is issued right AFTER the barrier that frees it
    ISSUE(0);
    if (nt > 1) { ISSUE(1); }
    if (nt > 2) { ISSUE(2); }
    if (nt > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // stage 0 landed
    for (int it = 0; it < nt; ++it) {
        // stage it + 1 must have landed before the consumers pass barrier `it`; stage it + 2 may still fly
        if (it + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // every read of stage `it` is done: its slot is free
        if (it + 3 < nt) { ISSUE(it + 3); }
    }
#endif
#undef ISSUE
}

template <int LAY>
__device__ __forceinline__ bf16x8 frag(const char* plane, int off0, int off1)
{
    if constexpr (LAY == ROW) {
        return *reinterpret_cast<const bf16x8*>(plane + off0);
    } else {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off1));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int LA, int LB, int AP, int LW>
__global__ __launch_bounds__(64 * (4 + LW)) void planes_gemm_kernel(PArgs g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qq = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.tiles_m <= g.tiles_n) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = ks * g.kchunk;
    const int nt = g.kchunk / 32;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    if (wave >= 4) {
        loader<LA, LB, AP, LW>(g, smem, wave - 4, lane, m0, n0, kbeg, nt);
        return;
    }
    unsigned long long t_in = 0, rt_in = 0;
    if (g.dbg) { t_in = __builtin_amdgcn_s_memtime(); rt_in = __builtin_amdgcn_s_memrealtime(); }
    const int r = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // fragment byte offsets inside a plane image, [32-row block][k16 step][first / second half (COL only)]
    int offA[2][2][2], offB[2][2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            {
                const int row = wm + 32 * b + r;
                if (LA == ROW) {
                    offA[b][s][0] = row * 64 + ((((2 * s + h) ^ ((row >> 2) & 3))) << 4);
                    offA[b][s][1] = 0;
                } else {
                    const int gq = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
                    const int ch = (wm + 32 * b) / 8 + 2 * gq + (p >> 1);
                    for (int tt = 0; tt < 2; ++tt) {
                        const int k = 16 * s + 8 * h + 4 * tt + q;
                        offA[b][s][tt] = k * 256 + ((ch ^ (q << 2)) << 4) + 8 * (p & 1);
                    }
                }
            }
            {
                const int row = wn + 32 * b + r;
                if (LB == ROW) {
                    offB[b][s][0] = row * 64 + ((((2 * s + h) ^ ((row >> 2) & 3))) << 4);
                    offB[b][s][1] = 0;
                } else {
                    const int gq = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
                    const int ch = (wn + 32 * b) / 8 + 2 * gq + (p >> 1);
                    for (int tt = 0; tt < 2; ++tt) {
                        const int k = 16 * s + 8 * h + 4 * tt + q;
                        offB[b][s][tt] = k * 256 + ((ch ^ (q << 2)) << 4) + 8 * (p & 1);
                    }
                }
            }
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    bf16x8 f0a[3][2], f0b[3][2], f1a[3][2], f1b[3][2];
    if (ABL == 3 || ABL == 4) {
        for (int pl = 0; pl < 3; ++pl) for (int a = 0; a < 2; ++a) {
            const s16x8 z = {(short)lane, 1, 2, 3, 4, 5, 6, (short)pl};
            f0a[pl][a] = f0b[pl][a] = f1a[pl][a] = f1b[pl][a] = __builtin_bit_cast(bf16x8, z);
        }
    }
#define FRAGS(FA, FB, BASE, S)                                                                \
    if (ABL != 3 && ABL != 4) {                                                               \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                          \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                       \
            if (pl < AP) FA[pl][a] = frag<LA>((BASE) + pl * PLANE, offA[a][S][0], offA[a][S][1]); \
            FB[pl][a] = frag<LB>((BASE) + (3 + pl) * PLANE, offB[a][S][0], offB[a][S][1]);     \
            if (ABL == 2) { if (pl < AP) asm volatile("" :: "v"(FA[pl][a])); asm volatile("" :: "v"(FB[pl][a])); } \
        } }
#define MMA(FA, FB)                                                                           \
    if (ABL != 2 && ABL != 4)                                                                 \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                             \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                       \
            if constexpr (AP == 3) {                                                          \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2][a], FB[0][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[1][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][a], FB[0][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], acc[a][b], 0, 0, 0); \
            } else {                                                                          \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[2][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[1][b], acc[a][b], 0, 0, 0); \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][a], FB[0][b], acc[a][b], 0, 0, 0); \
            }                                                                                 \
        }
    // DS-read instructions of one FRAGS block and MFMAs of one MMA block (for the issue-order directives)
    constexpr int NRD = 2 * (AP * (LA == COL ? 2 : 1) + 3 * (LB == COL ? 2 : 1)), NMM = 4 * (AP == 3 ? 6 : 3);
#if SCHED == 1
#define ORDER()                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < (NRD < NMM ? NRD : NMM); ++i_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
    }                                                                                         \
    if (NRD > NMM) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMM, 0);                 \
    if (NMM > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);
#define MID() do { } while (0)
#elif SCHED == 2
#define ORDER() do { } while (0)
#define MID() __builtin_amdgcn_sched_barrier(0)
#else
#define ORDER() do { } while (0)
#define MID() do { } while (0)
#endif
    __syncthreads();                                 // stage 0 landed
    unsigned long long t_loop = 0;
    if (g.dbg) t_loop = __builtin_amdgcn_s_memtime();
    FRAGS(f0a, f0b, smem, 0);
    for (int it = 0; it < nt; ++it) {
        const char* base = smem + (it % NSTAGE) * STAGE;
        const char* next = smem + ((it + 1) % NSTAGE) * STAGE;
        FRAGS(f1a, f1b, base, 1);
        MID();
        MMA(f0a, f0b);
        ORDER();
        __syncthreads();                             // all reads of stage `it` done; stage it + 1 landed
        FRAGS(f0a, f0b, next, 0);
        MID();
        MMA(f1a, f1b);
        ORDER();
    }
#undef FRAGS
#undef MMA
    if (g.dbg && wave == 0 && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), rt_end = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = g.dbg + 4 * (int64_t)blockIdx.x;
        d[0] = t_loop - t_in; d[1] = t_end - t_loop; d[2] = t_end - t_in; d[3] = rt_end - rt_in;
    }
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
}


// ---- the same GEMM on v_mfma_f32_16x16x32_bf16: one MFMA spans the whole 32-deep stage; per wave 4 x 4 tiles of 16 x 16.
// A fragments of a stage (12) are held for the whole stage and ping-pong across stages; B fragments come in two
// halves (column blocks 0-1, 2-3), the next half / the next stage's A prefetched under the current MFMAs.
typedef float f32x4a __attribute__((ext_vector_type(4)));
template <int LA, int LB, int AP, int LW>
__global__ __launch_bounds__(64 * (4 + LW)) void planes_gemm16_kernel(PArgs g)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qq = nwg >> 3, rem = nwg & 7;
    const int w = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + slot;
    const int tiles = g.tiles_m * g.tiles_n;
    const int ks = w / tiles, t = w - ks * tiles;
    int tm, tn;
    if (g.tiles_m <= g.tiles_n) { tn = t / g.tiles_m; tm = t - tn * g.tiles_m; }
    else { tm = t / g.tiles_n; tn = t - tm * g.tiles_n; }
    const int m0 = tm * 128, n0 = tn * 128;
    const int kbeg = ks * g.kchunk;
    const int nt = g.kchunk / 32;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= 4) {
        loader<LA, LB, AP, LW, 16>(g, smem, wave - 4, lane, m0, n0, kbeg, nt);
        return;
    }
    unsigned long long t_in = 0, rt_in = 0;
    if (g.dbg) { t_in = __builtin_amdgcn_s_memtime(); rt_in = __builtin_amdgcn_s_memrealtime(); }
    const int c16 = lane & 15, q = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    int offA[4][2], offB[4][2];                 // [16-row block][first / second tr read]
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        {
            const int row = wm + 16 * b + c16;
            if (LA == ROW) { offA[b][0] = row * 64 + ((q ^ row_swz(row, 16)) << 4); offA[b][1] = 0; }
            else {
                const int ch = (wm + 16 * b) / 8 + ((c16 & 3) >> 1);
                for (int tt = 0; tt < 2; ++tt) {
                    const int k = 8 * q + 4 * tt + (c16 >> 2);
                    offA[b][tt] = k * 256 + ((ch ^ col_swz(k, 16)) << 4) + 8 * (c16 & 1);
                }
            }
        }
        {
            const int row = wn + 16 * b + c16;
            if (LB == ROW) { offB[b][0] = row * 64 + ((q ^ row_swz(row, 16)) << 4); offB[b][1] = 0; }
            else {
                const int ch = (wn + 16 * b) / 8 + ((c16 & 3) >> 1);
                for (int tt = 0; tt < 2; ++tt) {
                    const int k = 8 * q + 4 * tt + (c16 >> 2);
                    offB[b][tt] = k * 256 + ((ch ^ col_swz(k, 16)) << 4) + 8 * (c16 & 1);
                }
            }
        }
    }
    f32x4a acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4a{0.f, 0.f, 0.f, 0.f};
    // Four fragment register groups, one per operand half (two 16-row blocks x planes): 96 VGPRs.  The stage's 4 x 4
    // tiles are visited quarter by quarter in a serpentine (lo,lo) (lo,hi) (hi,hi) (hi,lo) | (lo,hi) (lo,lo) (hi,lo) (hi,hi)
    // so that consecutive quarters share one half, and every quarter prefetches exactly ONE half (this stage's, or
    // after the mid-stage barrier the next stage's) under its 24 MFMAs.
    bf16x8 Alo[3][2], Ahi[3][2], Blo[3][2], Bhi[3][2];
#define RD_A(FA, BASE, HALF)                                                                  \
    _Pragma("unroll") for (int pl = 0; pl < AP; ++pl)                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
            FA[pl][i] = frag<LA>((BASE) + pl * PLANE, offA[2 * (HALF) + i][0], offA[2 * (HALF) + i][1]);
#define RD_B(FB, BASE, HALF)                                                                  \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
            FB[pl][i] = frag<LB>((BASE) + (3 + pl) * PLANE, offB[2 * (HALF) + i][0], offB[2 * (HALF) + i][1]);
#define MMQ(FA, FB, AH, BH)                                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                             \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                       \
            f32x4a& c = acc[2 * (AH) + a][2 * (BH) + b];                                      \
            if constexpr (AP == 3) {                                                          \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[2][a], FB[0][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[2][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1][a], FB[1][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[1][a], FB[0][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[1][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[0][b], c, 0, 0, 0);  \
            } else {                                                                          \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[2][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[1][b], c, 0, 0, 0);  \
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[0][a], FB[0][b], c, 0, 0, 0);  \
            }                                                                                 \
        }
    constexpr int RA = 2 * AP * (LA == COL ? 2 : 1), RB = 6 * (LB == COL ? 2 : 1), NM = 4 * (AP == 3 ? 6 : 3);
#define ORD(NR)                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < ((NR) < NM ? (NR) : NM); ++i_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
    }                                                                                         \
    if ((NR) > NM) __builtin_amdgcn_sched_group_barrier(0x100, (NR) - NM, 0);                 \
    if (NM > (NR)) __builtin_amdgcn_sched_group_barrier(0x008, NM - (NR), 0);
    __syncthreads();                                 // stage 0 landed
    unsigned long long t_loop = 0;
    if (g.dbg) t_loop = __builtin_amdgcn_s_memtime();
    RD_A(Alo, smem, 0);
    RD_B(Blo, smem, 0);
    for (int it = 0; it < nt; it += 2) {
        {   // even stage: holds (Alo, Blo)
            const char* base = smem + (it % NSTAGE) * STAGE;
            const char* next = smem + ((it + 1) % NSTAGE) * STAGE;
            RD_B(Bhi, base, 1); MMQ(Alo, Blo, 0, 0); ORD(RB);
            RD_A(Ahi, base, 1); MMQ(Alo, Bhi, 0, 1); ORD(RA);
            __syncthreads();                         // every read of stage `it` is done; stage it + 1 has landed
            RD_A(Alo, next, 0); MMQ(Ahi, Bhi, 1, 1); ORD(RA);
            RD_B(Bhi, next, 1); MMQ(Ahi, Blo, 1, 0); ORD(RB);
        }
        if (it + 1 < nt) {   // odd stage: holds (Alo, Bhi)
            const char* base = smem + ((it + 1) % NSTAGE) * STAGE;
            const char* next = smem + ((it + 2) % NSTAGE) * STAGE;
            RD_B(Blo, base, 0); MMQ(Alo, Bhi, 0, 1); ORD(RB);
            RD_A(Ahi, base, 1); MMQ(Alo, Blo, 0, 0); ORD(RA);
            __syncthreads();
            RD_A(Alo, next, 0); MMQ(Ahi, Blo, 1, 0); ORD(RA);
            RD_B(Blo, next, 0); MMQ(Ahi, Bhi, 1, 1); ORD(RB);
        }
    }
#undef RD_A
#undef RD_B
#undef MMQ
#undef ORD
    if (g.dbg && wave == 0 && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), rt_end = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = g.dbg + 4 * (int64_t)blockIdx.x;
        d[0] = t_loop - t_in; d[1] = t_end - t_loop; d[2] = t_end - t_in; d[3] = rt_end - rt_in;
    }
    // accumulator (16x16): col = lane & 15, row = 4 * (lane >> 4) + e
    float* C = g.C + (int64_t)ks * g.slab_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                C[(int64_t)(m0 + wm + 16 * a + 4 * q + e) * g.ldc + n0 + wn + 16 * b + c16] = acc[a][b][e];
}

template <int LA, int LB, int AP, int LW>
static int launch16(const PArgs& g, hipStream_t s)
{
    auto kern = planes_gemm16_kernel<LA, LB, AP, LW>;
    static bool attr = false;
    constexpr int lds = NSTAGE * STAGE;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.splitk), dim3(64 * (4 + LW)), lds, s, g);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// exact 3-way truncation split of an f32 matrix into bf16 planes (same layout, same ld)
__global__ void split_kernel(const float* __restrict__ X, int64_t n, unsigned short* __restrict__ P, int64_t plane)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4n v = *reinterpret_cast<const f32x4n*>(X + i);
    unsigned short o[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[j];
        const unsigned ua = __builtin_bit_cast(unsigned, a);
        const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u);
        const unsigned va = __builtin_bit_cast(unsigned, ra);
        const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u);
        o[0][j] = ua >> 16; o[1][j] = va >> 16; o[2][j] = __builtin_bit_cast(unsigned, sa) >> 16;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = o[p][0] | ((unsigned)o[p][1] << 16);
        w.y = o[p][2] | ((unsigned)o[p][3] << 16);
        *reinterpret_cast<uint2*>(P + p * plane + i) = w;
    }
}

template <int LA, int LB, int AP, int LW>
static int launch(const PArgs& g, hipStream_t s)
{
    auto kern = planes_gemm_kernel<LA, LB, AP, LW>;
    static bool attr = false;
    constexpr int lds = NSTAGE * STAGE;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.splitk), dim3(64 * (4 + LW)), lds, s, g);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int exp_split(const float* X, int64_t rows, int64_t ld, unsigned short* P, void* stream)
{
    const int64_t n = rows * ld;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, n, P, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int exp_gemm(int la, int lb, int ap, int lw, const unsigned short* A, int64_t lda, int64_t pa,
                        const unsigned short* B, int64_t ldb, int64_t pb, float* C, int64_t ldc, int64_t slab_stride,
                        int M, int N, int K, int splitk, void* stream, unsigned long long* dbg)
{
    if (M % 128 || N % 128 || K % (32 * splitk)) return -1;
    PArgs g{A, lda, pa, B, ldb, pb, C, ldc, slab_stride, M, N, K, K / splitk, splitk, M / 128, N / 128, dbg};
    hipStream_t s = (hipStream_t)stream;
#define CASE16(LAV, LBV, APV) if (la == LAV && lb == LBV && ap == APV && lw == 104) return launch16<LAV, LBV, APV, 4>(g, s)
    CASE16(ROW, COL, 3); CASE16(ROW, COL, 1); CASE16(ROW, ROW, 3); CASE16(ROW, ROW, 1); CASE16(COL, COL, 3);
#undef CASE16
#define CASE(LAV, LBV, APV, LWV) if (la == LAV && lb == LBV && ap == APV && lw == LWV) return launch<LAV, LBV, APV, LWV>(g, s)
    CASE(ROW, COL, 3, 4); CASE(ROW, COL, 3, 8); CASE(ROW, COL, 1, 4); CASE(ROW, COL, 1, 8);
    CASE(ROW, ROW, 3, 4); CASE(ROW, ROW, 3, 8); CASE(ROW, ROW, 1, 4); CASE(ROW, ROW, 1, 8);
    CASE(COL, COL, 3, 4); CASE(COL, COL, 3, 8);
#undef CASE
    return -4;
}
