/* xoshiro256** (Blackman & Vigna), the generator SURVEY.md 8(d) names for the synthetic workloads: the raw 64-bit stream for mauvealigner_amd/synth.py.
   Workload generation only -- bench and test infrastructure, not part of the product library (built as libmauve_synth.so by the same Makefile). */
#include <stdint.h>
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
/* s: the four state words, advanced in place; out[n]: the next n outputs */
void xo_fill(uint64_t *s, uint64_t *out, int64_t n)
{
    uint64_t s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    for (int64_t i = 0; i < n; i++) {
        out[i] = rotl(s1 * 5, 7) * 9;
        const uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45);
    }
    s[0] = s0; s[1] = s1; s[2] = s2; s[3] = s3;
}
/* n uniform base codes 0..3 from the top two bits of n successive outputs */
void xo_fill_bases(uint64_t *s, uint8_t *out, int64_t n)
{
    uint64_t s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    for (int64_t i = 0; i < n; i++) {
        out[i] = (uint8_t)((rotl(s1 * 5, 7) * 9) >> 62);
        const uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45);
    }
    s[0] = s0; s[1] = s1; s[2] = s2; s[3] = s3;
}
