// Kernel (b): products in R_q = Z_q[X]/(X^n+1).
//
// NTT-friendly q (prime, 2n | q-1): LDS-resident negacyclic NTT.  The forward transform is
// Cooley-Tukey with psi-powers in bit-reversed order (natural in, bit-reversed out), the
// inverse is Gentleman-Sande (bit-reversed in, natural out); the psi twist is merged into the
// twiddles so there is no separate pre/post scaling pass, and the pointwise product is done in
// bit-reversed order.  Twiddle multiplications are Shoup multiplications (precomputed
// floor(w*2^32/q)), i.e. one mul_hi + two mul_lo per butterfly.
//
// Every other q: exact schoolbook product in LDS (qf_ring_sb_kernel) so that the reference's
// own parameter sets (q = 2399, 92897729, 512, 16 ...) run unchanged.
//
// One launch fuses the whole scheme operation (see MODE_*): operands are read from HBM once,
// results written once.
#include "qf_internal.h"
#include "qf_ring.h"

#define RING_THREADS 256

// ---- in-LDS transforms over P polynomials stored back to back -------------------------------
__device__ __forceinline__ void ntt_forward_lds(uint32_t *s, uint32_t n, uint32_t lg, uint32_t P,
                                                const uint32_t *__restrict__ fw,
                                                const uint32_t *__restrict__ fwp, uint32_t q) {
    const uint32_t half = n >> 1, total = P * half;
    uint32_t lt = lg - 1;  // log2(t)
    for (uint32_t m = 1; m < n; m <<= 1, lt--) {
        for (uint32_t idx = threadIdx.x; idx < total; idx += RING_THREADS) {
            uint32_t p = idx >> (lg - 1), i = idx & (half - 1);
            uint32_t g = i >> lt, j = i & ((1u << lt) - 1);
            uint32_t lo = p * n + ((g << (lt + 1)) | j), hi = lo + (1u << lt);
            uint32_t u = s[lo];
            uint32_t v = qf_mul_shoup(s[hi], fw[m + g], fwp[m + g], q);
            s[lo] = qf_addmod(u, v, q);
            s[hi] = qf_submod(u, v, q);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void ntt_inverse_lds(uint32_t *s, uint32_t n, uint32_t lg, uint32_t P,
                                                const uint32_t *__restrict__ iw,
                                                const uint32_t *__restrict__ iwp, uint32_t q) {
    const uint32_t half = n >> 1, total = P * half;
    uint32_t lt = 0;
    for (uint32_t m = half; m >= 1; m >>= 1, lt++) {
        for (uint32_t idx = threadIdx.x; idx < total; idx += RING_THREADS) {
            uint32_t p = idx >> (lg - 1), i = idx & (half - 1);
            uint32_t g = i >> lt, j = i & ((1u << lt) - 1);
            uint32_t lo = p * n + ((g << (lt + 1)) | j), hi = lo + (1u << lt);
            uint32_t u = s[lo], v = s[hi];
            s[lo] = qf_addmod(u, v, q);
            s[hi] = qf_mul_shoup(qf_submod(u, v, q), iw[m + g], iwp[m + g], q);
        }
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(RING_THREADS) void qf_ring_ntt_kernel(QfRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t n = a.n, lg = a.lg, q = a.mod.q;
    const uint32_t T = n / 2 < RING_THREADS ? n / 2 : RING_THREADS;  // threads per item
    const uint32_t P = RING_THREADS / T;                              // items per workgroup
    const uint32_t *fw = a.tw, *fwp = a.tw + n, *iw = a.tw + 2 * n, *iwp = a.tw + 3 * n;
    uint32_t *s0 = lds, *s1 = lds + P * n;
    const uint64_t item0 = (uint64_t)blockIdx.x * P;
    const uint32_t PN = P * n;
    const uint32_t mu_stride = (n + 7) / 8;

    auto item_of = [&](uint32_t e) -> uint64_t { return item0 + (e >> lg); };
    auto valid = [&](uint32_t e) -> bool { return item_of(e) < a.B; };

    if (MODE == MODE_POLYMUL) {
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) {
            bool ok = valid(e);
            s0[e] = ok ? qf_red_u64(a.in0[item0 * n + e], a.mod) : 0;
            s1[e] = ok ? qf_red_u64(a.in1[item0 * n + e], a.mod) : 0;
        }
        __syncthreads();
        ntt_forward_lds(s0, n, lg, 2 * P, fw, fwp, q);  // s0 and s1 are contiguous: 2P polys
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) s0[e] = qf_mulmod(s0[e], s1[e], a.mod);
        __syncthreads();
        ntt_inverse_lds(s0, n, lg, P, iw, iwp, q);
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            if (valid(e)) a.out0[item0 * n + e] = qf_mul_shoup(s0[e], a.ninv, a.ninv_p, q);
    } else if (MODE == MODE_FWD_ONLY) {
        // key preparation: out0 = NTT(in0), out1 = Shoup companions
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            s0[e] = valid(e) ? qf_red_u64(a.in0[item0 * n + e], a.mod) : 0;
        __syncthreads();
        ntt_forward_lds(s0, n, lg, P, fw, fwp, q);
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            if (valid(e)) {
                uint32_t w = s0[e];
                a.out0[item0 * n + e] = w;
                a.out1[item0 * n + e] = (uint32_t)((((uint64_t)w) << 32) / q);
            }
    } else if (MODE == MODE_RLPR_ENC) {
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            s0[e] = valid(e) ? qf_red_u64(a.in0[item0 * n + e], a.mod) : 0;
        __syncthreads();
        ntt_forward_lds(s0, n, lg, P, fw, fwp, q);  // r^
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            const uint32_t *kh = a.key + (size_t)which * n, *khp = a.key_p + (size_t)which * n;
            for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) {
                uint32_t k = e & (n - 1);
                s1[e] = qf_mul_shoup(s0[e], kh[k], khp[k], q);
            }
            __syncthreads();
            ntt_inverse_lds(s1, n, lg, P, iw, iwp, q);
            const uint32_t *noise = which ? a.in2 : a.in1;
            uint32_t *out = which ? a.out1 : a.out0;
            for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
                if (valid(e)) {
                    uint64_t g = item0 * n + e;
                    uint32_t k = e & (n - 1);
                    uint32_t val = qf_mul_shoup(s1[e], a.ninv, a.ninv_p, q);
                    val = qf_addmod(val, qf_red_u64(noise[g], a.mod), q);
                    if (which && ((a.mu[item_of(e) * mu_stride + (k >> 3)] >> (k & 7)) & 1))
                        val = qf_addmod(val, a.mod.qhalf, q);
                    out[g] = val;
                }
            __syncthreads();
        }
    } else if (MODE == MODE_RLPR_DEC) {
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            s0[e] = valid(e) ? qf_red_u64(a.in0[item0 * n + e], a.mod) : 0;
        __syncthreads();
        ntt_forward_lds(s0, n, lg, P, fw, fwp, q);
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) {
            uint32_t k = e & (n - 1);
            s0[e] = qf_mul_shoup(s0[e], a.key[k], a.key_p[k], q);
        }
        __syncthreads();
        ntt_inverse_lds(s0, n, lg, P, iw, iwp, q);
        uint8_t *bitbuf = (uint8_t *)s1;  // one byte per coefficient
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) {
            uint32_t bit = 0;
            if (valid(e)) {
                uint32_t su = qf_mul_shoup(s0[e], a.ninv, a.ninv_p, q);
                uint32_t w = qf_submod(qf_red_u64(a.in1[item0 * n + e], a.mod), su, q);
                bit = qf_ring_dist(w, a.mod.qhalf, q) < qf_ring_dist(w, 0, q) ? 1u : 0u;
            }
            bitbuf[e] = (uint8_t)bit;
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < P * mu_stride; t += RING_THREADS) {
            uint32_t p = t / mu_stride, byte = t % mu_stride;
            if (item0 + p >= a.B) continue;
            uint32_t v = 0;
            for (uint32_t i = 0; i < 8 && byte * 8 + i < n; i++) v |= (uint32_t)bitbuf[p * n + byte * 8 + i] << i;
            a.outb[(item0 + p) * mu_stride + byte] = (uint8_t)v;
        }
    } else if (MODE == MODE_FDH) {
        // t = INTT( sum_j a^_j o NTT(sigma_j mod q) )
        const int32_t *sig = (const int32_t *)a.in0;
        uint32_t accr[16];  // PN / RING_THREADS <= 4096/256
#pragma unroll
        for (int i = 0; i < 16; i++) accr[i] = 0;
#pragma unroll 1
        for (uint32_t j = 0; j < a.npoly; j++) {
            for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS) {
                uint64_t it = item_of(e);
                uint32_t k = e & (n - 1);
                s0[e] = it < a.B ? qf_red_i64((int64_t)sig[(it * a.npoly + j) * n + k], a.mod) : 0;
            }
            __syncthreads();
            ntt_forward_lds(s0, n, lg, P, fw, fwp, q);
            const uint32_t *kh = a.key + (size_t)j * n, *khp = a.key_p + (size_t)j * n;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t e = threadIdx.x + i * RING_THREADS;
                if (e < PN) {
                    uint32_t k = e & (n - 1);
                    accr[i] = qf_addmod(accr[i], qf_mul_shoup(s0[e], kh[k], khp[k], q), q);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t e = threadIdx.x + i * RING_THREADS;
            if (e < PN) s0[e] = accr[i];
        }
        __syncthreads();
        ntt_inverse_lds(s0, n, lg, P, iw, iwp, q);
        for (uint32_t e = threadIdx.x; e < PN; e += RING_THREADS)
            if (valid(e)) a.out0[item0 * n + e] = qf_mul_shoup(s0[e], a.ninv, a.ninv_p, q);
    }
}

// ---- schoolbook fallback: one workgroup per item --------------------------------------------
struct SbAcc { uint64_t plo, phi, nlo, nhi; };
__device__ __forceinline__ uint32_t sb_coeff(const uint32_t *x, const uint32_t *y, uint32_t n, uint32_t k,
                                             const QfMod &m) {
    SbAcc c = {0, 0, 0, 0};
    for (uint32_t i = 0; i <= k; i++) { uint64_t p = (uint64_t)x[i] * y[k - i]; c.plo += (uint32_t)p; c.phi += p >> 32; }
    for (uint32_t i = k + 1; i < n; i++) { uint64_t p = (uint64_t)x[i] * y[k + n - i]; c.nlo += (uint32_t)p; c.nhi += p >> 32; }
    uint32_t pos = qf_addmod(qf_mulmod(qf_red_u64(c.phi, m), m.r32, m), qf_red_u64(c.plo, m), m.q);
    uint32_t neg = qf_addmod(qf_mulmod(qf_red_u64(c.nhi, m), m.r32, m), qf_red_u64(c.nlo, m), m.q);
    return qf_submod(pos, neg, m.q);
}

template <int MODE>
__global__ __launch_bounds__(RING_THREADS) void qf_ring_sb_kernel(QfRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t n = a.n, q = a.mod.q;
    uint32_t *x = lds, *y = lds + n, *acc = lds + 2 * n;
    const uint64_t it = blockIdx.x;
    const uint32_t mu_stride = (n + 7) / 8;
    if (MODE == MODE_POLYMUL) {
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) {
            x[k] = qf_red_u64(a.in0[it * n + k], a.mod);
            y[k] = qf_red_u64(a.in1[it * n + k], a.mod);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) a.out0[it * n + k] = sb_coeff(x, y, n, k, a.mod);
    } else if (MODE == MODE_RLPR_ENC) {
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) x[k] = qf_red_u64(a.in0[it * n + k], a.mod);
        for (int which = 0; which < 2; which++) {
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) y[k] = a.key_coef[(size_t)which * n + k];
            __syncthreads();
            const uint32_t *noise = which ? a.in2 : a.in1;
            uint32_t *out = which ? a.out1 : a.out0;
            for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) {
                uint32_t val = qf_addmod(sb_coeff(y, x, n, k, a.mod), qf_red_u64(noise[it * n + k], a.mod), q);
                if (which && ((a.mu[it * mu_stride + (k >> 3)] >> (k & 7)) & 1)) val = qf_addmod(val, a.mod.qhalf, q);
                out[it * n + k] = val;
            }
        }
    } else if (MODE == MODE_RLPR_DEC) {
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) {
            x[k] = qf_red_u64(a.in0[it * n + k], a.mod);
            y[k] = a.key_coef[k];
        }
        __syncthreads();
        uint8_t *bitbuf = (uint8_t *)acc;
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) {
            uint32_t w = qf_submod(qf_red_u64(a.in1[it * n + k], a.mod), sb_coeff(y, x, n, k, a.mod), q);
            bitbuf[k] = qf_ring_dist(w, a.mod.qhalf, q) < qf_ring_dist(w, 0, q) ? 1 : 0;
        }
        __syncthreads();
        for (uint32_t byte = threadIdx.x; byte < mu_stride; byte += RING_THREADS) {
            uint32_t v = 0;
            for (uint32_t i = 0; i < 8 && byte * 8 + i < n; i++) v |= (uint32_t)bitbuf[byte * 8 + i] << i;
            a.outb[it * mu_stride + byte] = (uint8_t)v;
        }
    } else if (MODE == MODE_FDH) {
        const int32_t *sig = (const int32_t *)a.in0;
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) acc[k] = 0;
        for (uint32_t j = 0; j < a.npoly; j++) {
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) {
                x[k] = qf_red_i64((int64_t)sig[(it * a.npoly + j) * n + k], a.mod);
                y[k] = a.key_coef[(size_t)j * n + k];
            }
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) acc[k] = qf_addmod(acc[k], sb_coeff(y, x, n, k, a.mod), q);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += RING_THREADS) a.out0[it * n + k] = acc[k];
    }
}

// ---- FDH finish: norm check + equality, one wave per signature ------------------------------
__global__ __launch_bounds__(256) void qf_fdh_finish_kernel(const int32_t *__restrict__ sigma, uint32_t len,
                                                            const uint32_t *__restrict__ t,
                                                            const uint32_t *__restrict__ u, uint32_t n, QfMod mod,
                                                            uint64_t norm_bound, uint64_t B, uint8_t *__restrict__ ok) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t b = wave; b < B; b += nwaves) {
        const int32_t *s = sigma + b * len;
        uint64_t lo = 0, hi = 0;  // sum of squares as split sum (each square < 2^62)
        for (uint32_t i = lane; i < len; i += 64) {
            int64_t v = s[i];
            uint64_t sq = (uint64_t)(v * v);
            lo += (uint32_t)sq; hi += sq >> 32;
        }
        lo = qf_wave_sum_u64(lo); hi = qf_wave_sum_u64(hi);
        uint32_t neq = 0;
        for (uint32_t k = lane; k < n; k += 64) neq |= (t[b * n + k] != qf_red_u64(u[b * n + k], mod));
        uint64_t anyneq = __ballot(neq != 0);
        if (lane == 0) {
            // total = hi*2^32 + lo compared with norm_bound without overflow
            uint64_t hi2 = hi + (lo >> 32), lo2 = lo & 0xFFFFFFFFull;
            bool in_domain = hi2 < (1ull << 32) && ((hi2 << 32) | lo2) <= norm_bound;
            ok[b] = (uint8_t)(in_domain && anyneq == 0);
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------
static void fill_common(QfRingArgs &a, const qf_ring *ring, uint64_t B) {
    a.n = ring->n; a.lg = ring->lg; a.mod = ring->mod; a.tw = ring->tw;
    a.ninv = ring->ninv; a.ninv_p = ring->ninv_p; a.B = B;
    a.in0 = a.in1 = a.in2 = nullptr; a.mu = nullptr; a.key = a.key_p = a.key_coef = nullptr;
    a.npoly = 0; a.out0 = a.out1 = nullptr; a.outb = nullptr;
    a.one_p = (uint32_t)((1ull << 32) / ring->q);
    a.off32 = (uint32_t)((((1ull << 31) + ring->q - 1) / ring->q) * ring->q);
    a.u = nullptr; a.ok = nullptr; a.norm_bound = 0;
    a.qinv32 = 0; a.ninv_m = 0; a.ninv_m_p = 0;
    if (ring->q & 1) {
        uint32_t inv = ring->q;                       // Newton: q*inv == 1 mod 2^32 after 5 steps (q odd)
        for (int i = 0; i < 5; i++) inv *= 2u - ring->q * inv;
        a.qinv32 = inv;
        a.ninv_m = (uint32_t)(((uint64_t)ring->ninv << 32) % ring->q);
        a.ninv_m_p = (uint32_t)((((uint64_t)a.ninv_m) << 32) / ring->q);
    }
}
template <int MODE>
static int launch_ring(qf_ctx *ctx, const qf_ring *ring, const QfRingArgs &a) {
    if (a.B == 0) return QF_OK;
    const uint32_t n = ring->n;
    if (ring->wave) return qf_launch_ring_wave(ctx, ring, MODE, a);
    if (ring->ntt) {
        uint32_t T = n / 2 < RING_THREADS ? n / 2 : RING_THREADS, P = RING_THREADS / T;
        size_t lds = (size_t)2 * P * n * 4;
        uint64_t blocks = (a.B + P - 1) / P;
        QfTimer t(ctx, &ctx->ms_ntt);
        hipLaunchKernelGGL(qf_ring_ntt_kernel<MODE>, dim3((unsigned)blocks), dim3(RING_THREADS), lds, ctx->stream, a);
    } else {
        if (MODE == MODE_FWD_ONLY) return QF_OK;
        size_t lds = (size_t)3 * n * 4;
        QfTimer t(ctx, &ctx->ms_ntt);
        hipLaunchKernelGGL(qf_ring_sb_kernel<MODE>, dim3((unsigned)a.B), dim3(RING_THREADS), lds, ctx->stream, a);
    }
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

int qf_launch_ntt_forward_keys(qf_ctx *ctx, const qf_ring *ring, qf_rkey *rk) {
    if (!ring->ntt) return QF_OK;
    QfRingArgs a; fill_common(a, ring, rk->npoly);
    a.in0 = rk->coef; a.out0 = rk->hat; a.out1 = rk->hat_p;
    return launch_ring<MODE_FWD_ONLY>(ctx, ring, a);
}
int qf_launch_polymul(qf_ctx *ctx, const qf_ring *ring, const uint32_t *x, const uint32_t *y, uint64_t B, uint32_t *c) {
    QfRingArgs a; fill_common(a, ring, B);
    a.in0 = x; a.in1 = y; a.out0 = c;
    return launch_ring<MODE_POLYMUL>(ctx, ring, a);
}
int qf_launch_ringlpr_enc(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *pk, const uint32_t *r,
                          const uint32_t *e1, const uint32_t *e2, const uint8_t *mu, uint64_t B,
                          uint32_t *u, uint32_t *v) {
    QfRingArgs a; fill_common(a, ring, B);
    a.in0 = r; a.in1 = e1; a.in2 = e2; a.mu = mu; a.key = pk->hat; a.key_p = pk->hat_p; a.key_coef = pk->coef;
    a.npoly = pk->npoly; a.out0 = u; a.out1 = v;
    return launch_ring<MODE_RLPR_ENC>(ctx, ring, a);
}
int qf_launch_ringlpr_dec(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *sk, const uint32_t *u,
                          const uint32_t *v, uint64_t B, uint8_t *mu) {
    QfRingArgs a; fill_common(a, ring, B);
    a.in0 = u; a.in1 = v; a.key = sk->hat; a.key_p = sk->hat_p; a.key_coef = sk->coef; a.npoly = sk->npoly; a.outb = mu;
    return launch_ring<MODE_RLPR_DEC>(ctx, ring, a);
}
int qf_launch_fdh_verify(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *ak, const int32_t *sigma,
                         const uint32_t *u, uint64_t norm_bound, uint64_t B, uint32_t *t, uint8_t *ok) {
    if (B == 0) return QF_OK;
    if (ring->wave) {  // one fused kernel: transforms, norm, comparison
        QfRingArgs a; fill_common(a, ring, B);
        a.in0 = (const uint32_t *)sigma; a.key = ak->hat; a.npoly = ak->npoly; a.out0 = t;
        a.u = u; a.ok = ok; a.norm_bound = norm_bound;
        return qf_launch_ring_wave(ctx, ring, MODE_FDH, a);
    }
    uint32_t *tbuf = t;
    if (!tbuf) {
        int rc = qf_ws_reserve(ctx, (size_t)B * ring->n * 4);
        if (rc) return rc;
        tbuf = (uint32_t *)ctx->ws;
    }
    QfRingArgs a; fill_common(a, ring, B);
    a.in0 = (const uint32_t *)sigma; a.key = ak->hat; a.key_p = ak->hat_p; a.key_coef = ak->coef; a.npoly = ak->npoly;
    a.out0 = tbuf;
    int rc = launch_ring<MODE_FDH>(ctx, ring, a);
    if (rc) return rc;
    uint64_t blocks = (B + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(qf_fdh_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sigma,
                       ak->npoly * ring->n, tbuf, u, ring->n, ring->mod, norm_bound, B, ok);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
