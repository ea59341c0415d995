// Philox4x32-10 counter-based RNG shared by device kernels and the host twin.
//
// Replaces Theano's MRG_RandomStreams (reference src/rbm.py:40,92,210,237,656), whose
// stream layout cannot be reproduced; parity with the oracle is on identical uniforms.
//
//   counter = (col, global_row >> 2, draw, step)     key = (seed_lo, seed_hi ^ stream_id)
//   word    = philox(counter, key)[global_row & 3]
//   uniform = ((word >> 8) + 0.5) * 2^-24            open interval (0,1), exact in f32
//
// One Philox block therefore serves 4 consecutive rows of one column -- the shape one
// lane of a 32x32 MFMA accumulator (and one thread of the split-K epilogue) holds.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MDBN_HD __host__ __device__ __forceinline__
#else
#define MDBN_HD static inline
#endif

struct PhiloxKey {
    uint32_t k0, k1, step, draw;
    uint64_t row_offset;
};

MDBN_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

MDBN_HD float philox_u01(uint32_t w)
{
    return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

#define MDBN_NORMAL_BIT 0x80000000u
