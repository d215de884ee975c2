// Counter-based dropout stream shared by every kernel that draws or regenerates a mask (device code only).
//
//   keep(i)  <=>  u16(i) >= thr,   thr = round(p * 65536)
//   u16(i)   =  low (i even) / high (i odd) half of  fmix32( (uint32)(i >> 1) * PHI + k0 )
//
// One murmur3 finaliser per PAIR of consecutive elements (the hash was > 50 % of the VALU work of the attention
// kernels when every element had its own two-round hash).  The pre-mix state is linear in the pair index, so a
// kernel walking a row pays one integer add per pair: state(j + d) = state(j) + d * PHI.  k0 is a 32-bit mix of
// (seed, site, call) made on the host (make_drop_key); the backward pass regenerates identical masks from
// (key, index), no mask tensor exists.  The pair index is taken modulo 2^32: a stream repeats after 2^33 elements
// (the largest tensor here has 2e8).  p is quantised to 2^-16 (0.1 -> 0.100006).
#pragma once
#include "kernels.h"

namespace gg {

constexpr uint32_t DROP_PHI = 0x9E3779B1u;

// One xorshift - multiply - xorshift round.  The input is a Weyl sequence (pair * PHI + k0, PHI odd), so its high bits are
// already equidistributed; the round spreads every input bit over both 16-bit halves.  A full two-multiply murmur
// finaliser measured as ~40 % of the hash cost (v_mul_lo_u32 is a quarter-rate instruction) for no statistical benefit a
// 10 % Bernoulli mask can show (tests/test_engine_oracle_gpu.py::test_dropout_statistics_and_replicas).
__device__ __forceinline__ uint32_t drop_fmix32(uint32_t h) {
#ifdef GG_DROP_HASH_MURMUR
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
#else
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15;
#endif
    return h;
}
// The key a kernel works with: k0 re-mixed with the engine's device-side epoch word (see DropKey::epoch), once per thread
// at kernel entry (a scalar load + 5 scalar ALU ops).  Forward and backward kernels of a step read the same word.
__device__ __forceinline__ DropKey drop_live(DropKey k) {
    if (k.epoch) {
        // readfirstlane: the word is the same for every lane, and everything derived from k0 (row states, hash inputs) must
        // stay in scalar registers - a vector-loaded k0 doubled the VGPR count of the attention kernels
        uint32_t h = k.k0 + (uint32_t)__builtin_amdgcn_readfirstlane((int)*k.epoch) * 0x632BE5ABu;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        k.k0 = h;
    }
    k.k0 += k.post;
    return k;
}
// pre-mix state of element pair `pair` (= element index >> 1)
__device__ __forceinline__ uint32_t drop_state(const DropKey& k, uint64_t pair) { return (uint32_t)pair * DROP_PHI + k.k0; }
// the two 16-bit uniforms of a pair: low half -> even element, high half -> odd element
__device__ __forceinline__ uint32_t drop_bits(uint32_t state) { return drop_fmix32(state); }
__device__ __forceinline__ bool drop_keep_even(uint32_t bits, uint32_t thr) { return (bits << 16) >= (thr << 16); }
__device__ __forceinline__ bool drop_keep_odd(uint32_t bits, uint32_t thr) { return bits >= (thr << 16); }

// generic single element
__device__ __forceinline__ float drop_factor(const DropKey& k, uint64_t i, float keep_scale) {
    const uint32_t b = drop_bits(drop_state(k, i >> 1));
    const bool keep = (i & 1) ? drop_keep_odd(b, k.thr) : drop_keep_even(b, k.thr);
    return keep ? keep_scale : 0.f;
}
// four consecutive elements starting at an EVEN index i0: two hashes
__device__ __forceinline__ void drop_factor4(const DropKey& k, uint64_t i0, float keep_scale, float (&f)[4]) {
    const uint32_t s0 = drop_state(k, i0 >> 1);
    const uint32_t b0 = drop_bits(s0), b1 = drop_bits(s0 + DROP_PHI);
    f[0] = drop_keep_even(b0, k.thr) ? keep_scale : 0.f;
    f[1] = drop_keep_odd(b0, k.thr) ? keep_scale : 0.f;
    f[2] = drop_keep_even(b1, k.thr) ? keep_scale : 0.f;
    f[3] = drop_keep_odd(b1, k.thr) ? keep_scale : 0.f;
}
// row stride of the attention-probability stream: rows start at multiples of 4 so that a lane's 4-key groups pair up
__host__ __device__ __forceinline__ int drop_attn_ld(int S) { return (S + 3) & ~3; }

}  // namespace gg
