// One-launch CD-k step for LDS-resident layers (mdbn_small.hip): arguments, LDS layout, launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdbn_kernels.h"

namespace mdbn {

constexpr int SM_ROWS = 16;                 // minibatch rows per slab = M of v_mfma_f32_16x16x4_f32
constexpr int SM_NW = 8, SM_NT = 64 * SM_NW;
constexpr int SM_MAXS = 12;                 // 16x16 tiles of S one wave may own (48 accumulator registers; 16 made hipcc spill)
constexpr int SM_MAXTH = 8;                 // ... of which at most 8 along H (H <= 128)
constexpr int SM_MAXKS = 8;                 // K chunks of a pass with fewer output tiles than waves
constexpr int SM_MAX_LDS = 160 * 1024;
constexpr int SM_MAX_BLOCKS = 64;           // workgroups (= partials the finish kernel sums); more slabs: a workgroup loops

struct SmallLayout {
    int Vp, Hp;                 // V, H rounded up to whole 16-column tiles
    int ldw;                    // row pitch of W in LDS: a multiple of 4 with ldw / 4 odd -- float4 rows, and the transposed
                                // read of propdown (16 rows apart per lane) touches 16 different bank quads
    int ldx, ldhs;              // row pitch of the 16-row visible / hidden buffers
    int tiles_up, tiles_dn;     // output tiles of propup (H) / propdown (V)
    int tg_up;                  // tiles a wave computes together in propup (they share the A operand: one LDS read feeds tg
                                // MFMAs, and tg independent accumulator chains hide the MFMA latency); propdown: always 4
    int ks_up;                  // K chunks per tile group of propup (its few output tiles alone would leave waves idle);
                                // propdown has tiles enough and is never split
    int rt;                     // 16-row tiles of S a wave owns along V (x all tiles_up along H): rt * tiles_up <= SM_MAXS
    int pld;                    // row pitch of the partial tiles of propup
    int oW, oXa, oXb, oHs, oMl, oPart, oCsP, oCsN, oCsV, oHb, oVb, oRed, oSrc;   // offsets in floats
    int bytes;
};

constexpr int SM_TG = 4;

__host__ __device__ inline SmallLayout small_layout(int V, int H, bool gauss)
{
    SmallLayout L;
    L.Vp = (V + 15) & ~15; L.Hp = (H + 15) & ~15;
    L.ldw = (H + 3) & ~3;
    if (((L.ldw >> 2) & 1) == 0) L.ldw += 4;
    // visible buffers: pitch = 20 (mod 64) floats -- the row-per-lane read of propup (A operand) is conflict-free and the
    // column-per-lane read of the statistics pass (A^T) two-way at worst
    int padx = ((20 - (L.Vp & 63)) + 64) & 63;
    if (padx < 4) padx += 64;
    L.ldx = L.Vp + padx; L.ldhs = L.Hp + 4;
    L.tiles_up = L.Hp / 16; L.tiles_dn = L.Vp / 16;
    L.tg_up = L.tiles_up < SM_TG ? L.tiles_up : SM_TG;
    {
        const int groups = (L.tiles_up + L.tg_up - 1) / L.tg_up, ksteps = (V + 3) / 4;
        L.ks_up = 1;
        while (L.ks_up < SM_MAXKS && groups * L.ks_up * 2 <= SM_NW && 8 * L.ks_up <= ksteps) L.ks_up *= 2;   // a chunk keeps >= 4 k-steps
    }
    L.rt = (L.tiles_dn + SM_NW - 1) / SM_NW;
    L.pld = L.Hp + 4;
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.oW = take(V * L.ldw);
    L.oXa = take(SM_ROWS * L.ldx);
    L.oXb = gauss ? L.oXa : take(SM_ROWS * L.ldx);
    L.oHs = take(SM_ROWS * L.ldhs);
    L.oMl = take(SM_ROWS * L.ldhs);
    L.oPart = take(L.ks_up * SM_ROWS * L.pld);
    L.oCsP = take(4 * L.Hp); L.oCsN = take(4 * L.Hp); L.oCsV = take(L.Vp);
    L.oHb = take(L.Hp); L.oVb = take(L.Vp);
    L.oRed = take(16);
    L.oSrc = take(2 * SM_ROWS);          // the slab's source rows (int64)
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
    float* part_S;                                   // [blocks][V * ldh]
    float* posP; float* negP;                        // [blocks][ldh]: sum_rows ph, sum_rows -nh
    float* partV;                                    // [blocks][ldv]: sum_rows (v0 - nv)
    float* cost_partials;                            // [blocks]
    // inspection copies (keep != 0) and chain taps (NULL = off), as mdbn_cd_args
    float* V2; float* P2; float* hs; float* vs;
    float* trace_h; float* trace_v;
    unsigned long long* stamps;                      // diagnostics (mdbn_set_stamp_buffer): wall-clock stamps of workgroup 0's phases
};

struct SmallFinArgs {
    const float* part; int nparts; int64_t part_stride;      // S partials
    int64_t n4;                                               // V * ldh / 4
    float* S_out;                                             // do_upd == 0: the summed S goes here
    int do_upd;
    UpdEpi upd;
    FinArgs fin;
};

int small_blocks(int64_t B);
bool small_shape_ok(int64_t B, int64_t V, int64_t H, int gauss);
hipError_t launch_small_cd(const SmallCdArgs& a, hipStream_t s);
hipError_t launch_small_finish(const SmallFinArgs& f, hipStream_t s);

}  // namespace mdbn
