// Kernel (b), fast path: negacyclic NTT with ONE WAVE PER POLYNOMIAL, register blocked.
//
// n = 2^LG (256..2048): a lane holds R = n/64 coefficients in registers.  A transform is NP =
// ceil(LG/RB) phases (RB = LG-6 register bits); in phase P the R registers of a lane cover RB
// index bits, so RB radix-2 stages run entirely in registers, with no memory traffic and no
// barrier.  Between phases the wave transposes its polynomial through a private LDS buffer
// (intra-wave: no s_barrier anywhere after the twiddle preload).  n = 1024: 4 + 4 + 2 stages.
//
//   phase 0: registers = top RB index bits  -> idx = lane + 64 r: HBM loads/stores are whole
//            256-byte runs per wave instruction, twiddles are wave-uniform (scalar loads);
//   middle : registers = next RB bits, lane = the rest; per-lane twiddles from the LDS copy;
//   last   : registers = the low remaining bits (+ top bits as filler).
//
// The forward transform is Cooley-Tukey (natural in, bit-reversed out), the inverse Gentleman-
// Sande; psi powers are merged into the twiddles (same table as qf_ntt.hip: psi^bitrev(i)), the
// pointwise stage runs in the last phase's register order, so no permutation is ever applied.
// Butterflies are Harvey's lazy form when q < 2^30 (values in [0,4q) forward, [0,2q) inverse:
// 3 multiplies + 6 adds/mins); otherwise every value is kept canonical.  All outputs are
// canonical residues, bit-identical to the general kernel and the oracle.
//
// Global accesses are written (base + lane)[64 * r]: pointer + compile-time offset folds into the
// instruction's immediate field.  Indexed as base[lane + 64 * r] with a 32-bit unsigned lane the sum
// may wrap, so hipcc kept one address register per r alive across the whole item loop (16 VGPRs per
// array, spilled in the FDH kernel).
//
// LDS: twiddle pairs (w, floor(w 2^32/q)) for both directions (16 KiB at n = 1024) + one
// n-dword transpose buffer per wave; the transposes use an XOR swizzle of the low five address
// bits (tools/ntt_swizzle_search.py) so that every ds_read_b32/ds_write_b32 is conflict free.
#include "qf_internal.h"
#include "qf_ring.h"

#define WV_THREADS 256
#define WV_WAVES 4

// QF_STAMP (diagnostic build, tools/ntt_stamp.py): per-wave s_memtime sums of the POLYMUL path -- global loads
// (issue + wait + range reduction), the two forward transforms, the pointwise product, the inverse transform,
// the scaling + stores, and, inside the transforms, the LDS transposes on their own.
#ifdef QF_STAMP
#define WV_T(var_) unsigned long long var_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var_) :: "memory"); __builtin_amdgcn_sched_barrier(0);
#define WV_STP_PARAM , unsigned long long *stp_tr
#define WV_STP_ARG , stp_tr
static unsigned long long *g_wv_dbg = nullptr;
extern "C" void qf_debug_set_buffer_ring(void *p) { g_wv_dbg = (unsigned long long *)p; }
#else
#define WV_T(var_)
#define WV_STP_PARAM
#define WV_STP_ARG
#endif

// ---- layouts ---------------------------------------------------------------------------------
template <int LG, int P>
struct Ph {
    static constexpr int RB = LG - 6;
    static constexpr int R = 1 << RB;
    static constexpr int raw = LG - (P + 1) * RB;
    static constexpr bool split = raw < 0;
    static constexpr int lo = split ? 0 : raw;
    static constexpr int nst = split ? LG - P * RB : RB;  // stages done in this phase
    // register bit k -> index bit
    static constexpr int regbit(int k) { return k < nst ? lo + k : LG - (RB - nst) + (k - nst); }
    static constexpr uint32_t cpart(int r) {
        uint32_t v = 0;
        for (int k = 0; k < RB; k++)
            if ((r >> k) & 1) v |= 1u << regbit(k);
        return v;
    }
    static __device__ __forceinline__ uint32_t lpart(uint32_t lane) {
        if (split) return lane << nst;
        return (lane & ((1u << lo) - 1u)) | ((lane >> lo) << (lo + RB));
    }
};

// XOR masks for index bits 5..11 (tools/ntt_swizzle_search.py)
template <int LG> struct Swz;
template <> struct Swz<8>  { static constexpr uint32_t m[7] = {6, 31, 1, 0, 0, 0, 0}; };
template <> struct Swz<9>  { static constexpr uint32_t m[7] = {6, 10, 25, 23, 0, 0, 0}; };
template <> struct Swz<10> { static constexpr uint32_t m[7] = {26, 31, 22, 26, 22, 0, 0}; };
template <> struct Swz<11> { static constexpr uint32_t m[7] = {7, 21, 27, 12, 19, 18, 0}; };
template <int LG>
__host__ __device__ constexpr uint32_t qf_swz(uint32_t i) {
    uint32_t x = 0;
    for (int b = 5; b < LG; b++)
        if ((i >> b) & 1) x ^= Swz<LG>::m[b - 5];
    return i ^ x;
}

// ---- butterflies -----------------------------------------------------------------------------
// low 32 bits of a*b + c in ONE instruction (v_mad_u64_u32; the upper half of its 64-bit addend is left
// undefined: only the low word of the result is used).  Shoup's  hi*q - Y*w  is then mul_hi, mul_lo, mad
// instead of mul_hi, mul_lo, mul_lo, sub -- with the NEGATED twiddle -w (forward table, negated when it is
// copied into LDS / read as a scalar) or the negated modulus (everything of the form x*w - hi*q).
#ifndef QF_NTT_MAD
#define QF_NTT_MAD 0   // measured (profiles/r3/ntt_ablation.log): polymul -1.5 %, RingLPR enc -2 %, dec spills at 4 waves/SIMD, FDH +0.5 %
#endif
typedef uint32_t wv_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wv_mad_lo(uint32_t a, uint32_t b_uniform, uint32_t c_lo) {
#if QF_NTT_MAD
    wv_u2 c, r;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
    c.x = c_lo;
#pragma clang diagnostic pop
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c) : "vcc");
    return r.x;
#else
    return a * b_uniform + c_lo;
#endif
}
// x*w - floor(x*wp / 2^32)*q  (Shoup, result in [0,2q) for w < q), nq = -q mod 2^32
__device__ __forceinline__ uint32_t wv_shoup(uint32_t x, uint32_t w, uint32_t wp, uint32_t q, uint32_t nq) {
#if QF_NTT_MAD
    return wv_mad_lo(__umulhi(x, wp), nq, x * w);
#else
    return x * w - __umulhi(x, wp) * q;
#endif
}
template <bool LAZY>
__device__ __forceinline__ void bf_fwd(uint32_t &X, uint32_t &Y, uint32_t w, uint32_t wp, uint32_t q, uint32_t q2) {
    if (LAZY) {
        // tn = -(Y*w mod q, lazy) mod 2^32: keeping the NEGATED product turns x + 2q - t into one
        // three-operand add (5 simple ops per butterfly instead of 6).  With QF_NTT_MAD the caller passes
        // w NEGATED (mod 2^32): tn = hi*q + Y*(-w)
#if QF_NTT_MAD
        uint32_t tn = wv_mad_lo(__umulhi(Y, wp), q, Y * w);
#else
        uint32_t tn = __umulhi(Y, wp) * q - Y * w;
#endif
        uint32_t x = min(X, X - q2);           // [0,4q) -> [0,2q)
        X = x - tn;
        Y = x + q2 + tn;
        return;
    }
    uint32_t t = Y * w - __umulhi(Y, wp) * q;  // [0, 2q) for any Y < 2^32
    {
        t = min(t, t - q);
        uint32_t x = X, s = x + t, d = x - t;
        X = min(s, s - q);
        Y = min(d, d + q);
    }
}
template <bool LAZY>
__device__ __forceinline__ void bf_inv(uint32_t &X, uint32_t &Y, uint32_t w, uint32_t wp, uint32_t q, uint32_t q2) {
    if (LAZY) {  // inputs in [0,2q)
        uint32_t s = X + Y, d = X + q2 - Y;
        X = min(s, s - q2);
        Y = wv_shoup(d, w, wp, q, 0u - q);
    } else {
        uint32_t s = X + Y, d = X - Y;
        X = min(s, s - q);
        d = min(d, d + q);
        uint32_t t = d * w - __umulhi(d, wp) * q;
        Y = min(t, t - q);
    }
}

template <int LG, bool LAZY>
struct Wv {
    static constexpr int N = 1 << LG, RB = LG - 6, R = 1 << RB, NP = (LG + RB - 1) / RB;

    // the RB (or fewer) stages of phase P on the registers of one lane
    template <int P, bool FWD>
    static __device__ __forceinline__ void stages(uint32_t (&x)[R], uint32_t lane, const uint2 *tw_lds,
                                                  const uint32_t *tw_g, uint32_t q) {
        using L = Ph<LG, P>;
        const uint32_t q2 = 2 * q;
        const uint32_t lp = L::lpart(lane);
        constexpr bool uniform = !L::split && L::lo >= 6;  // no lane bit above any stage bit
#pragma unroll
        for (int kk = 0; kk < L::nst; kk++) {
            const int k = FWD ? L::nst - 1 - kk : kk;
            const int b = L::regbit(k);
            const uint32_t m = 1u << (LG - 1 - b);
#pragma unroll
            for (int r = 0; r < R; r++) {
                if ((r >> k) & 1) continue;
                const int hi = r | (1 << k);
                uint32_t w, wp;
                if (uniform) {
                    const uint32_t ti = m + (L::cpart(r) >> (b + 1));
                    w = tw_g[ti]; wp = tw_g[N + ti];
                    if (QF_NTT_MAD && FWD && LAZY) w = 0u - w;   // scalar: the lazy forward butterfly takes -w (the LDS copy is stored negated)
                } else {
#ifdef QF_ABL_NOTWLDS
                    w = tw_g[m]; wp = tw_g[N + m];
                    if (QF_NTT_MAD && FWD && LAZY) w = 0u - w;
#else
                    const uint2 t2 = tw_lds[m + ((lp | L::cpart(r)) >> (b + 1))];
                    w = t2.x; wp = t2.y;
#endif
                }
                if (FWD) bf_fwd<LAZY>(x[r], x[hi], w, wp, q, q2);
                else bf_inv<LAZY>(x[r], x[hi], w, wp, q, q2);
            }
        }
    }

    template <int PF, int PT>
    static __device__ __forceinline__ void transpose(uint32_t (&x)[R], uint32_t lane, uint32_t *buf WV_STP_PARAM) {
#ifdef QF_ABL_NOTRANSPOSE
        return;
#endif
        WV_T(tt0)
        const uint32_t wa = qf_swz<LG>(Ph<LG, PF>::lpart(lane)), ra = qf_swz<LG>(Ph<LG, PT>::lpart(lane));
#pragma unroll
        for (int r = 0; r < R; r++) buf[wa ^ qf_swz<LG>(Ph<LG, PF>::cpart(r))] = x[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = buf[ra ^ qf_swz<LG>(Ph<LG, PT>::cpart(r))];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef QF_STAMP
        WV_T(tt1)
        if (stp_tr) *stp_tr += tt1 - tt0;
#endif
    }

    // twf_last: the twiddle table as the LAST phase sees it (the same LDS array; a caller may pass a
    // pointer the compiler cannot see through to keep that phase's loads inside its loop)
    template <int P>
    static __device__ __forceinline__ void fwd_rec(uint32_t (&x)[R], uint32_t lane, uint32_t *buf, const uint2 *twf,
                                                   const uint2 *twf_last, const uint32_t *tw_g, uint32_t q WV_STP_PARAM) {
        stages<P, true>(x, lane, P + 1 < NP ? twf : twf_last, tw_g, q);
        if constexpr (P + 1 < NP) {
            transpose<P, P + 1>(x, lane, buf WV_STP_ARG);
            fwd_rec<P + 1>(x, lane, buf, twf, twf_last, tw_g, q WV_STP_ARG);
        }
    }
    template <int P>
    static __device__ __forceinline__ void inv_rec(uint32_t (&x)[R], uint32_t lane, uint32_t *buf, const uint2 *twi,
                                                   const uint32_t *tw_g, uint32_t q WV_STP_PARAM) {
        stages<P, false>(x, lane, twi, tw_g, q);
        if constexpr (P > 0) {
            transpose<P, P - 1>(x, lane, buf WV_STP_ARG);
            inv_rec<P - 1>(x, lane, buf, twi, tw_g, q WV_STP_ARG);
        }
    }
    // x: natural order (idx = lane + 64 r), any value < 4q (lazy) / canonical -> NTT values in the
    // last phase's register order, in [0,4q) (lazy) / canonical
    static __device__ __forceinline__ void forward(uint32_t (&x)[R], uint32_t lane, uint32_t *buf, const uint2 *twf,
                                                   const uint32_t *tw_g, uint32_t q, const uint2 *twf_last = nullptr
#ifdef QF_STAMP
                                                   , unsigned long long *stp_tr = nullptr
#endif
    ) {
        fwd_rec<0>(x, lane, buf, twf, twf_last ? twf_last : twf, tw_g, q WV_STP_ARG);
    }
    // x: last-phase order, [0,2q) (lazy) / canonical -> n * (inverse transform), natural order,
    // same ranges; the caller scales by n^-1
    static __device__ __forceinline__ void inverse(uint32_t (&x)[R], uint32_t lane, uint32_t *buf, const uint2 *twi,
                                                   const uint32_t *tw_g_inv, uint32_t q
#ifdef QF_STAMP
                                                   , unsigned long long *stp_tr = nullptr
#endif
    ) {
        inv_rec<NP - 1>(x, lane, buf, twi, tw_g_inv, q WV_STP_ARG);
    }
};

// x < 2^32 -> [0,2q) (one_p = floor(2^32/q)); canonical when CANON
template <bool CANON>
__device__ __forceinline__ uint32_t wv_red32(uint32_t x, uint32_t one_p, uint32_t q) {
    uint32_t r = x - __umulhi(x, one_p) * q;
    return CANON ? min(r, r - q) : r;
}
// any uint32 -> canonical residue.  q >= 2^30 (the non-lazy kernels): x < 2^32 <= 4q, two conditional
// subtractions (4 simple operations instead of mul_hi + mul_lo + 3)
template <bool LAZY>
__device__ __forceinline__ uint32_t wv_canon(uint32_t x, uint32_t one_p, uint32_t q) {
    if (LAZY) return wv_red32<true>(x, one_p, q);
    x = min(x, x - 2 * q);
    return min(x, x - q);
}

// waves per SIMD the register allocator must leave room for (n = 1024 only; measured choices,
// tools/time_ring.py over tools/build_variant.sh builds)
#ifndef QF_WV_W_POLYMUL
#define QF_WV_W_POLYMUL 3
#endif
#ifndef QF_WV_W_FDH
#define QF_WV_W_FDH 2
#endif
#ifndef QF_WV_W_ENC
#define QF_WV_W_ENC 2
#endif
#ifndef QF_WV_FDH_HOIST
#define QF_WV_FDH_HOIST 2   // 0: no twiddles kept across the transforms of a signature, 1: the middle phase's, 2: all
#endif
template <int LG, int MODE>
constexpr int wv_min_waves() {
    if (LG != 10) return 1;
    return MODE == MODE_RLPR_ENC ? QF_WV_W_ENC : (MODE == MODE_FDH ? QF_WV_W_FDH : (MODE == MODE_POLYMUL ? QF_WV_W_POLYMUL : 4));
}

template <int LG, int MODE, bool LAZY>
__global__ __launch_bounds__(WV_THREADS, (wv_min_waves<LG, MODE>())) void qf_ring_wave_kernel(QfRingArgs a) {
    using W = Wv<LG, LAZY>;
    constexpr int N = W::N, R = W::R;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint2 *twf = (uint2 *)lds, *twi = twf + N;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *buf = lds + 4 * N + wave * N;
    for (uint32_t e = tid; e < (uint32_t)N; e += WV_THREADS) {
        twf[e] = make_uint2((QF_NTT_MAD && LAZY) ? 0u - a.tw[e] : a.tw[e], a.tw[N + e]);   // lazy forward butterflies take -w (bf_fwd)
        twi[e] = make_uint2(a.tw[2 * N + e], a.tw[3 * N + e]);
    }
    __syncthreads();
    const uint32_t q = a.mod.q, q2 = 2 * q, one_p = a.one_p, nq = 0u - q;
    const uint32_t *twg_f = a.tw, *twg_i = a.tw + 2 * N;
    const uint32_t LI = Ph<LG, W::NP - 1>::lpart(lane);  // unused except for documentation of order
    (void)LI;

    // canonical result of an inverse-transform output in the key-based modes: the prepared keys carry
    // the factor n^-1 (MODE_FWD_ONLY), so nothing is left to scale -- [0,2q) (lazy) -> [0,q)
    auto finish = [&](uint32_t v) -> uint32_t { return LAZY ? min(v, v - q) : v; };
    auto load_nat = [&](uint32_t(&x)[R], const uint32_t *src) {
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = (src + lane)[64 * r];
        // the first forward stage needs its "X" inputs below 4q (lazy) / everything canonical.  Lazy with
        // q >= 2^29: any uint32 is below 8q, one conditional subtraction of 4q does it
        if (LAZY && q >= (1u << 29)) {
#pragma unroll
            for (int r = 0; r < R / 2; r++) x[r] = min(x[r], x[r] - 4 * q);
        } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (LAZY) { if (r < R / 2) x[r] = wv_red32<false>(x[r], one_p, q); }
                else x[r] = wv_canon<false>(x[r], one_p, q);
            }
        }
    };

    for (uint64_t item = (uint64_t)blockIdx.x * WV_WAVES + wave; item < a.B; item += (uint64_t)gridDim.x * WV_WAVES) {
        // RLPR_ENC: a compiler barrier per item keeps hipcc from holding every stage's twiddles live
        // across items (it did, at 256 VGPRs + AGPR copies and one wave per SIMD); re-read from LDS per
        // item the kernel fits two waves per SIMD without scratch
        if constexpr (MODE == MODE_RLPR_ENC) asm volatile("" ::: "memory");
        if constexpr (MODE == MODE_POLYMUL) {
            uint32_t xa[R], xb[R];
            WV_T(ts0)
            load_nat(xa, a.in0 + item * N);
            load_nat(xb, a.in1 + item * N);
            WV_T(ts1)
#ifdef QF_STAMP
            unsigned long long s_tr = 0;
            W::forward(xa, lane, buf, twf, twg_f, q, nullptr, &s_tr);
            W::forward(xb, lane, buf, twf, twg_f, q, nullptr, &s_tr);
#else
            W::forward(xa, lane, buf, twf, twg_f, q);
            W::forward(xb, lane, buf, twf, twg_f, q);
#endif
            WV_T(ts2)
            // pointwise product in Montgomery form: a*b*2^-32 mod q with 4 multiplies (a 64-bit Barrett
            // product takes 10); the 2^-32 is absorbed by the final n^-1 scaling (ninv_m = n^-1 * 2^32).
            // a is canonicalised so that a*b < q*2^32 (b < 4q, q < 2^30 in the lazy variant).
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t va = xa[r];
                if (LAZY) { va = min(va, va - q2); va = min(va, va - q); }
                const uint32_t lo = va * xb[r], hi = __umulhi(va, xb[r]);
                const uint32_t v = hi - __umulhi(lo * a.qinv32, q) + q;   // (0, 2q)
                xa[r] = LAZY ? v : min(v, v - q);
            }
            WV_T(ts3)
#ifdef QF_STAMP
            W::inverse(xa, lane, buf, twi, twg_i, q, &s_tr);
#else
            W::inverse(xa, lane, buf, twi, twg_i, q);
#endif
            WV_T(ts4)
            uint32_t *dst = a.out0 + item * N;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t t = wv_shoup(xa[r], a.ninv_m, a.ninv_m_p, q, nq);
                (dst + lane)[64 * r] = min(t, t - q);
            }
#ifdef QF_STAMP
            WV_T(ts5)
            if (a.dbg && lane == 0) {
                unsigned long long *d = a.dbg + (((uint64_t)blockIdx.x * WV_WAVES + wave) * 8);
                d[0] += ts1 - ts0; d[1] += ts2 - ts1; d[2] += ts3 - ts2; d[3] += ts4 - ts3; d[4] += ts5 - ts4; d[5] += s_tr; d[6] += 1; d[7] += ts5 - ts0;
            }
#endif
        } else if constexpr (MODE == MODE_FWD_ONLY) {
            // key preparation: interleaved (value, Shoup companion) pairs in last-phase register order
            uint32_t x[R];
            load_nat(x, a.in0 + item * N);
            W::forward(x, lane, buf, twf, twg_f, q);
            uint2 *dst = (uint2 *)a.out0 + item * N;
#pragma unroll
            for (int r = 0; r < R; r++) {
                // n^-1 goes into the key: the key-based modes then need no scaling after their inverse transforms
                uint32_t v = wv_shoup(x[r], a.ninv, a.ninv_p, q, nq);
                v = min(v, v - q);
                (dst + lane)[64 * r] = make_uint2(v, (uint32_t)((((uint64_t)v) << 32) / q));
            }
        } else if constexpr (MODE == MODE_RLPR_ENC) {
            // the noise polynomials are fetched ahead of the transforms that precede their use
            // (e1 before the forward transform, e2 before the first inverse), so their HBM latency
            // hides behind the butterflies instead of stalling the three resident waves
            uint32_t xr[R], t[R], e1[R], e2[R];
            const uint32_t *n1 = a.in1 + item * N, *n2 = a.in2 + item * N;
#pragma unroll
            for (int r = 0; r < R; r++) e1[r] = (n1 + lane)[64 * r];
            load_nat(xr, a.in0 + item * N);
            // message bits: word j of the row holds coefficients 32j..32j+31
            const uint32_t *mrow = (const uint32_t *)(a.mu + item * (uint64_t)(N / 8));
            const uint32_t mword = mrow[lane < (uint32_t)(N / 32) ? lane : 0];
            W::forward(xr, lane, buf, twf, twg_f, q);
            const uint2 *kha = (const uint2 *)a.key, *khb = (const uint2 *)a.key + N;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint2 k2 = (kha + lane)[64 * r];
                uint32_t v = wv_shoup(xr[r], k2.x, k2.y, q, nq);  // [0,2q)
                t[r] = LAZY ? v : min(v, v - q);
            }
            constexpr bool PRE2 = R <= 16;  // n = 2048: 128 data registers would spill
            if (PRE2) {
#pragma unroll
                for (int r = 0; r < R; r++) e2[r] = (n2 + lane)[64 * r];
            }
            W::inverse(t, lane, buf, twi, twg_i, q);
            uint32_t *du = a.out0 + item * N, *dv = a.out1 + item * N;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t val = finish(t[r]) + wv_canon<LAZY>(e1[r], one_p, q);
                (du + lane)[64 * r] = min(val, val - q);
            }
            if (!PRE2) {
#pragma unroll
                for (int r = 0; r < R; r++) e2[r] = (n2 + lane)[64 * r];
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint2 k2 = (khb + lane)[64 * r];
                uint32_t v = wv_shoup(xr[r], k2.x, k2.y, q, nq);
                t[r] = LAZY ? v : min(v, v - q);
            }
            W::inverse(t, lane, buf, twi, twg_i, q);
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t val = finish(t[r]) + wv_canon<LAZY>(e2[r], one_p, q);
                val = min(val, val - q);
                uint32_t wlo = __builtin_amdgcn_readlane(mword, (2 * r) % (N / 32));
                uint32_t whi = __builtin_amdgcn_readlane(mword, (2 * r + 1) % (N / 32));
                uint32_t wsel = lane < 32 ? wlo : whi;
                val += ((wsel >> (lane & 31)) & 1u) ? a.mod.qhalf : 0u;
                (dv + lane)[64 * r] = min(val, val - q);
            }
        } else if constexpr (MODE == MODE_RLPR_DEC) {
            uint32_t x[R], vraw[R];
            const uint32_t *vv = a.in1 + item * N;
#pragma unroll
            for (int r = 0; r < R; r++) vraw[r] = (vv + lane)[64 * r];  // fetched ahead of the transforms
            load_nat(x, a.in0 + item * N);
            W::forward(x, lane, buf, twf, twg_f, q);
            const uint2 *kh = (const uint2 *)a.key;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint2 k2 = (kh + lane)[64 * r];
                uint32_t v = wv_shoup(x[r], k2.x, k2.y, q, nq);
                x[r] = LAZY ? v : min(v, v - q);
            }
            W::inverse(x, lane, buf, twi, twg_i, q);
            uint32_t myword = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t su = finish(x[r]);
                uint32_t w = qf_submod(wv_canon<LAZY>(vraw[r], one_p, q), su, q);
                bool bit = qf_ring_dist(w, a.mod.qhalf, q) < qf_ring_dist(w, 0, q);
                uint64_t m = __ballot(bit);  // coefficients 64r .. 64r+63
                if (lane == (uint32_t)(2 * r)) myword = (uint32_t)m;
                if (lane == (uint32_t)(2 * r + 1)) myword = (uint32_t)(m >> 32);
            }
            uint32_t *orow = (uint32_t *)(a.outb + item * (uint64_t)(N / 8));
            if (lane < (uint32_t)(N / 32)) orow[lane] = myword;
        } else if constexpr (MODE == MODE_FDH) {
            // t = INTT( sum_j a^_j o NTT(sigma_j mod q) ); ok = [||sigma||^2 <= bound] && [t == u]
            const int32_t *sig = (const int32_t *)a.in0 + item * (uint64_t)a.npoly * N;
            uint32_t acc[R], x[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0;
            uint64_t nlo = 0, nhi = 0;  // exact sum of squares, split so nothing overflows
#pragma unroll 1
            for (uint32_t j = 0; j < a.npoly; j++) {
                // hipcc hoists loop-invariant twiddle loads out of this loop: wanted for the middle phase (its
                // 15 (w, w') pairs stay in registers across the 32 transforms of a signature), but with the
                // last phase's 12 pairs too the kernel needs more than 256 VGPRs and drops to one wave per
                // SIMD.  The last phase therefore reads through a pointer the compiler cannot see through.
                const uint2 *twf_j = twf;
                if (QF_WV_FDH_HOIST < 2) asm volatile("" : "+v"(twf_j));
                const int32_t *sj = sig + (size_t)j * N;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    int32_t sv = (sj + lane)[64 * r];
                    uint64_t sq = (uint64_t)((int64_t)sv * (int64_t)sv);
                    nlo += (uint32_t)sq; nhi += sq >> 32;
                    // off32 = smallest multiple of q >= 2^31 lies in [2^31, 2^31 + q): added to NEGATIVE
                    // coefficients only (sv + off32 in [0, off32)); a non-negative sv is its own
                    // 32-bit representative (adding off32 to it could wrap past 2^32)
                    const uint32_t sr = (uint32_t)sv + (a.off32 & (uint32_t)(sv >> 31));
                    x[r] = LAZY ? wv_red32<false>(sr, one_p, q) : wv_canon<false>(sr, one_p, q);
                }
                W::forward(x, lane, buf, QF_WV_FDH_HOIST == 0 ? twf_j : twf, twg_f, q, twf_j);
                // the 16 key pairs are fetched AFTER the transform (they are L2-resident; fetched ahead
                // they would hold 32 more registers through it)
                asm volatile("" ::: "memory");
                const uint2 *kh = (const uint2 *)a.key + (size_t)j * N;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    uint2 k2 = (kh + lane)[64 * r];
                    uint32_t v = wv_shoup(x[r], k2.x, k2.y, q, nq);  // [0,2q)
                    if (LAZY) { uint32_t s = acc[r] + v; acc[r] = min(s, s - q2); }
                    else { v = min(v, v - q); uint32_t s = acc[r] + v; acc[r] = min(s, s - q); }
                }
            }
            W::inverse(acc, lane, buf, twi, twg_i, q);
            const uint32_t *urow = a.u + item * N;
            uint32_t neq = 0;
#pragma unroll
            for (int r = 0; r < R; r++) {
                acc[r] = finish(acc[r]);
                neq |= (acc[r] != wv_canon<LAZY>((urow + lane)[64 * r], one_p, q));
            }
            if (a.out0) {   // one uniform branch around all the stores (not one per coefficient)
                uint32_t *trow = a.out0 + item * N + lane;
#pragma unroll
                for (int r = 0; r < R; r++) trow[64 * r] = acc[r];
            }
            nlo = qf_wave_sum_u64(nlo); nhi = qf_wave_sum_u64(nhi);
            const uint64_t anyneq = __ballot(neq != 0);
            if (lane == 0) {
                uint64_t hi2 = nhi + (nlo >> 32), lo2 = nlo & 0xFFFFFFFFull;
                bool in_domain = hi2 < (1ull << 32) && ((hi2 << 32) | lo2) <= a.norm_bound;
                a.ok[item] = (uint8_t)(in_domain && anyneq == 0);
            }
        }
    }
}

// ---- launcher --------------------------------------------------------------------------------
template <int LG, int MODE>
static int wv_launch(qf_ctx *ctx, const qf_ring *ring, const QfRingArgs &a_in) {
    constexpr int N = 1 << LG;
    QfRingArgs a = a_in;
#ifdef QF_STAMP
    a.dbg = g_wv_dbg;
#endif
    const size_t lds = (size_t)(4 * N + WV_WAVES * N) * 4;
    uint64_t blocks = (a.B + WV_WAVES - 1) / WV_WAVES;
    if (blocks > 256 * 8) blocks = 256 * 8;  // persistent beyond that: waves loop over items
    const bool lazy = ring->q < (1u << 30);
    QfTimer t(ctx, &ctx->ms_ntt);
    if (lazy) {
        static QfPerDevice attr;
        QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_ring_wave_kernel<LG, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
        hipLaunchKernelGGL((qf_ring_wave_kernel<LG, MODE, true>), dim3((unsigned)blocks), dim3(WV_THREADS), lds, ctx->stream, a);
    } else {
        static QfPerDevice attr;
        QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_ring_wave_kernel<LG, MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
        hipLaunchKernelGGL((qf_ring_wave_kernel<LG, MODE, false>), dim3((unsigned)blocks), dim3(WV_THREADS), lds, ctx->stream, a);
    }
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

template <int MODE>
static int wv_dispatch(qf_ctx *ctx, const qf_ring *ring, const QfRingArgs &a) {
    switch (ring->lg) {
        case 8: return wv_launch<8, MODE>(ctx, ring, a);
        case 9: return wv_launch<9, MODE>(ctx, ring, a);
        case 10: return wv_launch<10, MODE>(ctx, ring, a);
        case 11: return wv_launch<11, MODE>(ctx, ring, a);
    }
    return qf_fail(ctx, QF_ERR_NOTSUP, "wave NTT: degree not instantiated");
}

bool qf_ring_wave_supported(uint32_t lg) { return lg >= 8 && lg <= 11; }

int qf_launch_ring_wave(qf_ctx *ctx, const qf_ring *ring, int mode, const QfRingArgs &a) {
    if (a.B == 0) return QF_OK;
    switch (mode) {
        case MODE_POLYMUL: return wv_dispatch<MODE_POLYMUL>(ctx, ring, a);
        case MODE_RLPR_ENC: return wv_dispatch<MODE_RLPR_ENC>(ctx, ring, a);
        case MODE_RLPR_DEC: return wv_dispatch<MODE_RLPR_DEC>(ctx, ring, a);
        case MODE_FDH: return wv_dispatch<MODE_FDH>(ctx, ring, a);
        case MODE_FWD_ONLY: return wv_dispatch<MODE_FWD_ONLY>(ctx, ring, a);
    }
    return qf_fail(ctx, QF_ERR_ARG, "wave NTT: bad mode");
}
