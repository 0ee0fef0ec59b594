// Device-side helpers shared by all gfx950 kernels: modular reduction, the seeded generator,
// signed radix-2^8 digits.  Host-side structs live in qf_internal.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Modulus descriptor passed by value to kernels.
struct QfMod {
    uint32_t q;
    uint32_t qhalf;  // floor(q/2)
    uint64_t mu;     // floor(2^64 / q)
    uint64_t off;    // smallest multiple of q that is >= 2^62
    uint32_t r32;    // 2^32 mod q
    uint32_t pad;
};

// x in [0, 2^64) -> x mod q.  Barrett with mu = floor(2^64/q): the estimate is at most one
// short, so a single conditional subtraction canonicalises (2q < 2^32 keeps it in 32 bits).
__device__ __forceinline__ uint32_t qf_red_u64(uint64_t x, const QfMod &m) {
    uint64_t qh = __umul64hi(x, m.mu);
    uint32_t r = (uint32_t)x - (uint32_t)qh * m.q;
    return r >= m.q ? r - m.q : r;
}
// |x| <= 2^62 (signed) -> canonical residue
__device__ __forceinline__ uint32_t qf_red_i64(int64_t x, const QfMod &m) {
    return qf_red_u64((uint64_t)(x + (int64_t)m.off), m);
}
__device__ __forceinline__ uint32_t qf_addmod(uint32_t a, uint32_t b, uint32_t q) {
    uint32_t s = a + b;  // a,b < q < 2^31: no wrap
    return s >= q ? s - q : s;
}
__device__ __forceinline__ uint32_t qf_submod(uint32_t a, uint32_t b, uint32_t q) {
    return a >= b ? a - b : a + q - b;
}
__device__ __forceinline__ uint32_t qf_mulmod(uint32_t a, uint32_t b, const QfMod &m) {
    return qf_red_u64((uint64_t)a * b, m);
}
// Shoup multiplication by a constant w with companion wp = floor(w * 2^32 / q); x < 2^32 any.
// Result in [0, 2q) (lazy) -- callers canonicalise when they need to.
__device__ __forceinline__ uint32_t qf_mul_shoup_lazy(uint32_t x, uint32_t w, uint32_t wp, uint32_t q) {
    uint32_t t = __umulhi(x, wp);
    return x * w - t * q;
}
__device__ __forceinline__ uint32_t qf_mul_shoup(uint32_t x, uint32_t w, uint32_t wp, uint32_t q) {
    uint32_t r = qf_mul_shoup_lazy(x, w, wp, q);
    return r >= q ? r - q : r;
}
// ring (wrap-around) distance and the reference's decision rule (regev.rs:445-451)
__device__ __forceinline__ uint32_t qf_ring_dist(uint32_t a, uint32_t x, uint32_t q) {
    uint32_t d = a >= x ? a - x : x - a;
    return min(d, q - d);
}
__device__ __forceinline__ uint32_t qf_decide_bit(uint32_t r, const QfMod &m) {
    return qf_ring_dist(r, 0, m.q) > qf_ring_dist(r, m.qhalf, m.q) ? 1u : 0u;
}

// ---- seeded generator (include/qfgpu.h "seeded generator"; oracle/qf_oracle.c sm64) ----
__device__ __forceinline__ uint64_t qf_sm64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t qf_rand64(uint64_t base, uint64_t idx) { return qf_sm64(base + idx); }
__device__ __forceinline__ uint32_t qf_rand_below(uint64_t base, uint64_t idx, uint32_t bound) {
    return (uint32_t)__umul64hi(qf_rand64(base, idx), (uint64_t)bound);
}

// ---- the generator behind every qf_sample_* entry point: SplitMix64 in counter mode under a 64-bit
// seed (tests, benches, oracle parity), or -- the *_keyed_dev entry points, used whenever a host object
// was constructed WITHOUT a seed -- ChaCha20 in counter mode under a 256-bit key: value(idx) = the first
// 8 bytes (little-endian) of the ChaCha20 block with block counter idx (64-bit: state words 12, 13) and
// nonce `base` (state words 14, 15).  A public matrix drawn from it reveals nothing about the secret
// streams of the same key (SplitMix64's finaliser is a bijection of a 64-bit state: not a CSPRNG).
struct QfGen {
    uint64_t base;      // SplitMix64: seed ^ (stream << 48); ChaCha20: the 64-bit nonce, nonce ^ (stream << 48)
    uint32_t key[8];
    uint32_t keyed;
    uint32_t pad;
};
#define QF_CHACHA_QR(a, b, c, d)                                         \
    a += b; d ^= a; d = (d << 16) | (d >> 16); c += d; b ^= c; b = (b << 12) | (b >> 20); \
    a += b; d ^= a; d = (d << 8) | (d >> 24);  c += d; b ^= c; b = (b << 7) | (b >> 25);
__device__ __forceinline__ uint64_t qf_chacha64(const QfGen &g, uint64_t idx) {
    const uint32_t c0 = 0x61707865u, c1 = 0x3320646eu, c2 = 0x79622d32u, c3 = 0x6b206574u;
    uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = g.key[0], x5 = g.key[1], x6 = g.key[2], x7 = g.key[3],
             x8 = g.key[4], x9 = g.key[5], x10 = g.key[6], x11 = g.key[7], x12 = (uint32_t)idx, x13 = (uint32_t)(idx >> 32),
             x14 = (uint32_t)g.base, x15 = (uint32_t)(g.base >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        QF_CHACHA_QR(x0, x4, x8, x12) QF_CHACHA_QR(x1, x5, x9, x13) QF_CHACHA_QR(x2, x6, x10, x14) QF_CHACHA_QR(x3, x7, x11, x15)
        QF_CHACHA_QR(x0, x5, x10, x15) QF_CHACHA_QR(x1, x6, x11, x12) QF_CHACHA_QR(x2, x7, x8, x13) QF_CHACHA_QR(x3, x4, x9, x14)
    }
    return (uint64_t)(x0 + c0) | ((uint64_t)(x1 + c1) << 32);
}
__device__ __forceinline__ uint64_t qf_gen64(const QfGen &g, uint64_t idx) {
    return g.keyed ? qf_chacha64(g, idx) : qf_sm64(g.base + idx);   // wave-uniform branch
}
__device__ __forceinline__ uint32_t qf_gen_below(const QfGen &g, uint64_t idx, uint32_t bound) {
    return (uint32_t)__umul64hi(qf_gen64(g, idx), (uint64_t)bound);
}

// 64-bit wave reductions (wave = 64 lanes)
__device__ __forceinline__ uint64_t qf_wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
