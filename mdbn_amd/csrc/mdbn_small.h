// One-launch CD-k step for LDS-resident layers (mdbn_small.hip): arguments, LDS layout, launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdbn_kernels.h"

namespace mdbn {

constexpr int SM_ROWS = 4;                  // minibatch rows per slab = M of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 columns),
                                            // and the rows one Philox block serves
constexpr int SM_NW = 8, SM_NT = 64 * SM_NW;    // the waves that run the passes (a ninth draws the random numbers: mdbn_small.hip)
constexpr int SM_MAXQ = 16;                 // accumulators of the statistics per wave: one 64-row x 64-column tile of S
constexpr int SM_MAX_LDS = 160 * 1024;
constexpr int SM_MAX_BLOCKS = 128;          // workgroups (= partials the finish kernel sums); more slabs: a workgroup loops

struct SmallLayout {
    int Vp, Hp;                 // V, H rounded up to 16: one A register feeds 16 k-steps (block broadcast, see mdbn_small.hip)
    int V64, H64;               // ... to whole 64-column tiles (one MFMA covers 64 output columns)
    int ldw;                    // row pitch of W in LDS: a multiple of 4 with ldw / 4 odd -- float4 rows, and the transposed
                                // read of propdown (one row per lane, 16 bytes each) touches 16 different bank quads
    int ldx, ldhs;              // row pitch of the 4-row visible / hidden buffers: = 8 (mod 32), so that the A-register read
                                // (lane 4 b + i reads row i, column k0 + b) meets 32 different banks per half wave
    int tiles_up, tiles_dn;     // 64-column output tiles of propup (H) / propdown (V)
    int ks_up, per_up;          // propup: K chunks (its one or two output tiles alone would leave waves idle) of per_up k-steps
    int oW, oX0, oXa, oXb, oHs, oM0, oMn, oPart, oCsP, oCsN, oCsV, oHb, oVb, oU;   // offsets in floats
    int bytes;
};

__host__ __device__ inline SmallLayout small_layout(int V, int H, bool gauss)
{
    SmallLayout L;
    L.Vp = (V + 15) & ~15; L.Hp = (H + 15) & ~15;
    L.V64 = (V + 63) & ~63; L.H64 = (H + 63) & ~63;
    // a few fixed pitches for the usual widths (propup's loop is compiled for each: immediate offsets instead of address adds)
    if (H <= 20) L.ldw = 20;
    else if (H <= 44) L.ldw = 44;
    else if (H <= 68) L.ldw = 68;
    else if (H <= 132) L.ldw = 132;
    else {
        L.ldw = (H + 3) & ~3;
        if (((L.ldw >> 2) & 1) == 0) L.ldw += 4;
    }
    L.ldx = L.V64 + 8; L.ldhs = L.H64 + 8;
    L.tiles_up = L.H64 / 64; L.tiles_dn = L.V64 / 64;
    {
        int ks = SM_NW / L.tiles_up;
        if (ks < 1) ks = 1;
        const int groups = L.Vp / 16;
        L.per_up = 16 * ((groups + ks - 1) / ks);
        L.ks_up = (L.Vp + L.per_up - 1) / L.per_up;
    }
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.oW = take(L.Vp * L.ldw + 16);      // (+ slack: the last row's 16-wide k group reads past the pitch)
    L.oX0 = take(SM_ROWS * L.ldx);
    L.oXa = take(SM_ROWS * L.ldx);
    L.oXb = gauss ? L.oXa : take(SM_ROWS * L.ldx);
    L.oHs = take(SM_ROWS * L.ldhs);
    L.oM0 = take(SM_ROWS * L.ldhs);
    L.oMn = take(SM_ROWS * L.ldhs);
    L.oPart = take(L.ks_up * L.H64 * 4);
    L.oCsP = take(L.H64); L.oCsN = take(L.H64); L.oCsV = take(L.V64);
    L.oHb = take(L.H64); L.oVb = take(L.V64);
    L.oU = take(4 * L.H64);
    L.bytes = o * 4;
    return L;
}

struct SmallCdArgs {
    const float* data; int64_t n_data, ld_data;      // the training matrix
    const void* idx; int idx64;                      // [B] minibatch rows, or NULL (rows 0..B-1)
    int B, V, H, k, gauss, keep;
    int64_t ldv, ldh;                                // leading dimensions of the [., V] / [., H] outputs; W is [V][ldh]
    const float* W; const float* hbias; const float* vbias;
    PhiloxKey rng;                                   // .draw unused (the step numbers its own draws)
    SmallLayout L;                                   // LDS layout (small_layout; filled in by launch_small_cd)
    // one partial per workgroup
    float* part_S;                                   // [blocks][small_part_quads] float4, in the lanes' order (mdbn_small.hip)
    float* posP; float* negP;                        // [blocks][ldh]: sum_rows ph, sum_rows -nh
    float* partV;                                    // [blocks][ldv]: sum_rows (v0 - nv)
    float* cost_partials;                            // [blocks][SM_NW]: one per wave
    // inspection copies (keep != 0) and chain taps (NULL = off), as mdbn_cd_args
    float* V2; float* P2; float* hs; float* vs;
    float* trace_h; float* trace_v;
    unsigned long long* stamps;                      // diagnostics (mdbn_set_stamp_buffer): wall-clock stamps of workgroup 0's phases
};

// 16-byte pieces of one workgroup's S partial: ldh / 4 pieces for each of the V64 lanes (pad lanes hold zeros)
__host__ __device__ inline int small_part_quads(const SmallLayout& L, int ldh) { return (ldh >> 2) * L.V64; }

struct SmallFinArgs {
    const float* part; int nparts;                            // S partials, small_part_quads float4 each, in the lanes' order
    int64_t n4p;                                              // small_part_quads
    int V, q4, tiles_dn;                                      // rows of S, ldh / 4, V64 / 64: to decode a position
    int lanes;                                                // threads that share the sum of one float4 of S (1, 2, 4, 8 or 16)
    float* S_out;                                             // do_upd == 0: the summed S goes here ([V][ldh])
    int do_upd;
    UpdEpi upd;
    FinArgs fin;
};

extern int g_small_fin_lanes;
int small_blocks(int64_t B);
bool small_shape_ok(int64_t B, int64_t V, int64_t H, int gauss);
bool small_ld_ok(int64_t V, int64_t H, int64_t ldv, int64_t ldh);       // leading dimensions the LDS images can take
hipError_t launch_small_cd(const SmallCdArgs& a, hipStream_t s);
hipError_t launch_small_finish(const SmallFinArgs& f, hipStream_t s);

}  // namespace mdbn
