/* C twin of the device RNG (Philox4x32-10) -- TEST INFRASTRUCTURE, see oracle/__init__.py.
 *
 * Second, independent restatement of oracle/philox_np.py used to cross-check the numpy
 * twin and the HIP kernels bit for bit.  The reference's own randomness is Theano's
 * MRG_RandomStreams (rbm.py:40,92,210,237,656), not reproducible offline; parity is
 * defined on identical uniforms.  Built into oracle/_build/libphilox_ref.so by
 * oracle/Makefile (gcc only).
 */
#include <stdint.h>
#include <stddef.h>

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* raw block, for the Random123 known-answer vectors */
void philox_ref_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

/* uniforms in (0,1) addressed exactly as oracle/philox_np.py documents */
void philox_ref_uniform(float *out, int64_t rows, int64_t cols, int64_t ld,
                        uint64_t seed, uint32_t stream, uint32_t step, uint32_t draw,
                        uint64_t row_offset)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ stream;
    for (int64_t r = 0; r < rows; ++r) {
        uint64_t g = (uint64_t)r + row_offset;
        for (int64_t j = 0; j < cols; ++j) {
            uint32_t c[4] = {(uint32_t)j, (uint32_t)(g >> 2), draw, step};
            philox4x32_10(c, k0, k1);
            uint32_t w = c[g & 3];
            out[r * ld + j] = ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f);
        }
    }
}
