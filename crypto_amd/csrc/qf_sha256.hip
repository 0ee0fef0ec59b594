// Hash-to-Z_q / hash-to-R_q for FDH verification: the reference's
// hash_to_mat_zq_sha256 (construction/hash/sha256.rs:110-133) for B messages at once.
//
// Entry (i,j) of the rows x cols target of message s is
//     int( sha256_hex("0 {q} {i} {j} {rows} {cols} {s}"), 16 ) mod q
// (sha256.rs:65-83 nests the strings at :70,79,120,127; the chunk counter only reaches 1 for
// moduli of 128 bits and more, and q < 2^31 here, so there is exactly one digest per entry).
// The ring target of HashMatPolynomialRingZq with rows = cols = 1 (sha256.rs:249-256,
// signature/fdh/gpv_ring.rs:78-82) is the (n x 1) case: coefficient r = entry (r, 0).
//
// One thread per entry: it formats its decimal prefix into a private LDS line, streams
// prefix || message through the SHA-256 compression function (message schedule in registers,
// rounds fully unrolled), and folds the 256-bit big-endian digest mod q word by word.
#include "qf_internal.h"

#define SH_THREADS 256
#define SH_PRE 68  // bytes of LDS per thread: 57-byte prefix at most, 17-dword stride (no bank conflicts)

__constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

__device__ __forceinline__ void sha_compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        if (t >= 16) {
            const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
            const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[t & 15] += s0 + w[(t + 9) & 15] + s1;
        }
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = h + S1 + ch + SHA_K[t] + w[t & 15];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// decimal digits of v into p[0..], returns the count
__device__ __forceinline__ uint32_t put_dec(uint8_t *p, uint32_t v) {
    // 32-bit only: the division by 10 is a multiply-high, and the digit count comes from compares
    const uint32_t n = 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) +
                       (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
    for (uint32_t i = n; i-- > 0;) { const uint32_t d = v / 10u; p[i] = (uint8_t)('0' + (v - d * 10u)); v = d; }
    return n;
}

__global__ __launch_bounds__(SH_THREADS) void qf_hash_to_zq_kernel(const uint8_t *__restrict__ msgs,
                                                                   const uint64_t *__restrict__ offs, uint64_t B,
                                                                   uint32_t rows, uint32_t cols, QfMod mod,
                                                                   uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t pre_all[SH_THREADS * SH_PRE];
    uint8_t *pre = pre_all + threadIdx.x * SH_PRE;
    const uint64_t E = (uint64_t)rows * cols;
    const uint64_t idx = (uint64_t)blockIdx.x * SH_THREADS + threadIdx.x;
    if (idx >= B * E) return;
    const uint64_t b = idx / E, e = idx % E;
    const uint32_t i = (uint32_t)(e / cols), j = (uint32_t)(e % cols);
    // "0 {q} {i} {j} {rows} {cols} "
    uint32_t pl = 0;
    pre[pl++] = '0'; pre[pl++] = ' ';
    pl += put_dec(pre + pl, mod.q); pre[pl++] = ' ';
    pl += put_dec(pre + pl, i); pre[pl++] = ' ';
    pl += put_dec(pre + pl, j); pre[pl++] = ' ';
    pl += put_dec(pre + pl, rows); pre[pl++] = ' ';
    pl += put_dec(pre + pl, cols); pre[pl++] = ' ';
    const uint8_t *m = msgs + offs[b];
    const uint64_t ml = offs[b + 1] - offs[b];
    const uint64_t total = pl + ml;                    // bytes hashed
    const uint64_t nblk = (total + 9 + 63) / 64;
    const uint64_t bitlen = total * 8;
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto byte_at = [&](uint64_t p) -> uint32_t {
        if (p < pl) return pre[p];
        if (p < total) return m[p - pl];
        return p == total ? 0x80u : 0u;
    };
    for (uint64_t blk = 0; blk < nblk; blk++) {
        uint32_t w[16];
        const uint64_t base = blk * 64;
        if (base >= pl && base + 64 <= total) {
            // block entirely inside the message: no per-byte case analysis
            const uint8_t *mp = m + (base - pl);
#pragma unroll
            for (int t = 0; t < 16; t++)
                w[t] = ((uint32_t)mp[4 * t] << 24) | ((uint32_t)mp[4 * t + 1] << 16) | ((uint32_t)mp[4 * t + 2] << 8) | mp[4 * t + 3];
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint64_t p = base + 4 * t;
                uint32_t wv;
                if (p + 4 <= pl) {                        // inside the prefix: one aligned LDS word
                    wv = __builtin_bswap32(*(const uint32_t *)(pre + p));
                } else if (p >= pl && p + 4 <= total) {   // inside the message
                    const uint8_t *mp = m + (p - pl);
                    wv = ((uint32_t)mp[0] << 24) | ((uint32_t)mp[1] << 16) | ((uint32_t)mp[2] << 8) | mp[3];
                } else if (p > total) {                   // pure padding
                    wv = 0;
                } else {                                  // a word that straddles a boundary
                    wv = (byte_at(p) << 24) | (byte_at(p + 1) << 16) | (byte_at(p + 2) << 8) | byte_at(p + 3);
                }
                w[t] = wv;
            }
        }
        if (blk == nblk - 1) { w[14] = (uint32_t)(bitlen >> 32); w[15] = (uint32_t)bitlen; }
        sha_compress(st, w);
    }
    // 256-bit big-endian integer mod q
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r = qf_red_u64(((uint64_t)r << 32) | st[k], mod);
    out[idx] = r;
}

int qf_launch_hash_to_zq(qf_ctx *ctx, const uint8_t *msgs, const uint64_t *offs, uint64_t B, uint32_t rows,
                         uint32_t cols, const QfMod &mod, uint32_t *out) {
    const uint64_t total = B * (uint64_t)rows * cols;
    if (!total) return QF_OK;
    const uint64_t blocks = (total + SH_THREADS - 1) / SH_THREADS;
    if (blocks > 0x7fffffffull) return qf_fail(ctx, QF_ERR_NOTSUP, "hash: too many entries for one launch");
    hipLaunchKernelGGL(qf_hash_to_zq_kernel, dim3((unsigned)blocks), dim3(SH_THREADS), 0, ctx->stream, msgs, offs, B,
                       rows, cols, mod, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
