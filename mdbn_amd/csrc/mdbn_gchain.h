// Group-chain CD-k (mdbn_gchain.hip): arguments, geometry, launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdbn_kernels.h"
#include "mdbn_thin.h"

namespace mdbn {

constexpr int GC_NT = 512;
constexpr int GC_MAX_LDS = 160 * 1024;
constexpr int GC_MAX_G = 16;                // members of a group: one-round exchange (every member reads all g partials)
constexpr int GC_SPIN_LIMIT = 4000000;      // polls (x s_sleep 8 ~ 0.2 us) before a member gives up: ~1 s
constexpr int GC_MAX_FLAGS = 2 * 256;       // flag words of a context: [groups][2][g], groups * g <= 256

struct GChainGeom {
    int g, Vb;                              // members per group; rows of W per member (a multiple of 16)
    int nslab, nsg;                         // 32-row slabs; groups running at once
    int PW, S1;                             // LDS pitch of W; K shares of the visible pass
    int lds;                                // dynamic LDS (bytes)
    int64_t xbuf_floats;                    // exchange payload: [nsg][2][g][32][ldh]
};

// bytes of LDS for Vb rows per member (layout: gchain_kernel); S1 = K shares of the visible pass that still fit
__host__ __device__ inline int gchain_lds(int Vb, int ldh, bool gauss, int& S1)
{
    const int R16 = Vb, R32 = (Vb + 31) & ~31, K16 = (ldh + 15) & ~15;
    const int PW = thin_pitch(K16), PH = K16 + 8, PX = thin_pitch(R16);
    const int ntile1 = R32 >> 5;
    (void)gauss;
    const int fixed = R16 * PW * 4 + 32 * PX * 4 + (K16 + R16) * 4;      // W block, visible tile, biases
    const int hk = 32 * PH * 2;
    for (S1 = 8 / ntile1 > 0 ? 8 / ntile1 : 1; S1 >= 1; --S1) {
        const int red = S1 * 32 * R32 * 4;
        const int tot = fixed + (((hk > red ? hk : red) + 15) & ~15);
        if (tot <= GC_MAX_LDS) return tot;
    }
    S1 = 1;
    return fixed + (((hk > 32 * R32 * 4 ? hk : 32 * R32 * 4) + 15) & ~15);
}

// Does the group-chain step serve this shape?  More than one thin batch of rows, ldh <= 512, a block of W per member that
// fits one CU's LDS with g <= 16 members (and <= 8 tiles of 32 rows: one per wave in the visible pass), and layers big
// enough that the multi-launch chain is the cost (tiny layers are LDS-resident: the one-launch path takes them first).
__host__ __device__ inline bool gchain_geom(int64_t B, int64_t V, int64_t H, int64_t ldv, int64_t ldh, int gauss, int num_cu, GChainGeom& t)
{
    if (B <= TH_MAXB || B > 65536 || ldh > 512 || ldh % 4 || ldv % 4 || V < 32 || H < 1 || V > 8192) return false;
    for (int g = 2; g <= GC_MAX_G; g *= 2) {
        const int Vb = (int)(((V + g - 1) / g + 15) & ~int64_t(15));
        if (Vb > 256) continue;
        int S1 = 1;
        const int lds = gchain_lds(Vb, (int)ldh, gauss != 0, S1);
        if (lds > GC_MAX_LDS) continue;
        t.g = g; t.Vb = Vb; t.S1 = S1; t.lds = lds;
        t.PW = thin_pitch((int)((ldh + 15) & ~int64_t(15)));
        t.nslab = (int)((B + 31) / 32);
        const int cus = num_cu > 0 ? num_cu : 1;
        int nsg = cus / g;
        if (nsg * g > GC_MAX_FLAGS / 2) nsg = GC_MAX_FLAGS / 2 / g;
        if (nsg < 1) return false;
        t.nsg = nsg < t.nslab ? nsg : t.nslab;
        t.xbuf_floats = (int64_t)t.nsg * 2 * g * 32 * ldh;
        return true;
    }
    return false;
}

struct GChainArgs {
    int B, V, H, k, gauss;
    int64_t ldv, ldh;
    int g, Vb, nslab, nsg, PW, S1;
    const float* W; const float* hbias; const float* vbias;
    const float* data; int64_t n_data, ld_data;
    const void* idx; int idx64;
    float* V2; float* P2; float* hs; float* vs;
    float* trace_h; float* trace_v;
    float* colPpos; float* colPneg; float* colV;    // 4-row column partials of the bias statistics (as act_quad writes them)
    float* cost_partials;                           // [nslab * g]
    float* xbuf;                                    // exchange payload [nsg][2][g][32][ldh]
    unsigned* flags;                                // [nsg][2][g], owned by the context (only the exchange protocol writes them)
    unsigned* error;                                // += 1 per exchange a member gave up on
    unsigned seq0;                                  // sequence number of this launch's first exchange
    PhiloxKey rng;
};

hipError_t launch_gchain(const GChainArgs& a, int lds, hipStream_t s);

}  // namespace mdbn
