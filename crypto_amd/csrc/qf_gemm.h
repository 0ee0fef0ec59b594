// Launch arguments and helpers of the mod-q GEMM (qf_gemm.hip).
#pragma once
#include "qf_internal.h"

struct QfGemmArgs {
    const int8_t *kplanes;   // key digit planes, fragment order
    uint32_t rb;             // key row blocks
    uint32_t ks;             // k-steps (multiple of QF_KS)
    uint32_t rows;           // key rows (= output row length)
    const uint8_t *xbits;    // BITS: bit-packed rows
    size_t xstride;          // BITS: row stride in bytes (multiple of 16); k-tile-major bits: bytes between the planes of two k-tiles
    uint32_t xkt;            // BITS: 1 = k-tile-major (QF_X_BITS_KT): the 16 bytes of row b, k-tile t at t * xstride + b * 16
    const int8_t *rem_planes;// bit operands: packed remainder block of the key [ks][64][16], or null
    uint32_t rem_rows;       //   its rows (rows mod 32)
    const int8_t *xplanes;   // PLANES: batch digit planes, fragment order [mb][ks][xlb_total][64][16]
    uint32_t xlb_total;      //   digits stored per (message block, k-step)
    uint32_t xlb_off;        //   first digit this launch consumes (its LB digits are xlb_off .. xlb_off+LB-1)
    uint32_t mchunk;         // PLANES: message tiles per raster chunk (0 = plain grid order), see qf_gemm_body
    uint64_t B;              // messages
    const uint32_t *addend;  // B x rows or null
    const uint8_t *msg;      // B or null
    uint32_t *out;           // B x rows
    QfMod mod;
    int32_t cs[8];           // centred 2^(8s) mod q
    double qd, qinv;         // q and 1/q as doubles (fp64 epilogue of the one-digit kernels)
    uint32_t epi2;           //   1: reduce the upper digit pair before recombining (see qf_epi_f64)
    unsigned long long *dbg; // diagnostic build only (QF_STAMP): per-wave cycle sums
};

// 16 bits -> 16 bytes of {0,1}: each nibble is spread over one dword (bit t -> byte t)
__device__ __forceinline__ v4i qf_expand16(uint32_t w16) {
    v4i r;
    r.x = (int)((((w16)&0xFu) * 0x00204081u) & 0x01010101u);
    r.y = (int)((((w16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
    r.z = (int)((((w16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
    r.w = (int)((((w16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
    return r;
}

// Epilogue arithmetic of the one-digit-operand kernels (LB = 1: every BASELINE shape), in fp64: the
// int64 version costs 9 quarter-rate 64-bit multiply-adds per element (4 to recombine, 3 in the
// Barrett reduction, 2 for the address) and made the epilogue a third of the LPR launch.
//   x = a0 + 256 a1 + 65536 (a2 + 256 a3) + extra     exact in a double: the host guarantees
//       |x| < 2^52, or sets epi2 and the upper pair is reduced mod q first (|65536 hi'| < 2^47);
//   t = rint(x / q), r = x - t q                        one fma, exact: an integer, |r| <= q/2 + 1/4;
//   result = min(r, r + q) as unsigned words            (r < 0 wraps to >= 2^31 > r + q; q < 2^31).
// Signed digits represent an entry EXACTLY (sum d_s 256^s = its representative), so the true powers
// of 256 are used here, not the centred residues cs[].
// the same without addend / message term and with the one-step recombination known to be exact
// (Regev and SIS: most row blocks of most launches) -- 11 instructions
template <int S>
__device__ __forceinline__ uint32_t qf_epi_f64_plain(int a0, int a1, int a2, int a3, const QfGemmArgs &a) {
    static_assert(S >= 1 && S <= 4, "one-digit operands: S = LA <= 4");
    double x = (double)a0;
    if constexpr (S >= 2) x = __builtin_fma((double)a1, 256.0, x);
    if constexpr (S >= 3) {
        double hi = (double)a2;
        if constexpr (S >= 4) hi = __builtin_fma((double)a3, 256.0, hi);
        x = __builtin_fma(hi, 65536.0, x);
    }
    const double t = __builtin_rint(x * a.qinv);
    const uint32_t r = (uint32_t)(int)__builtin_fma(-t, a.qd, x);
    const uint32_t rq = r + a.mod.q;
    return r < rq ? r : rq;
}
template <int S>
__device__ __forceinline__ uint32_t qf_epi_f64(int a0, int a1, int a2, int a3, double extra, const QfGemmArgs &a) {
    static_assert(S >= 1 && S <= 4, "one-digit operands: S = LA <= 4");
    double x = (double)a0 + extra;
    if constexpr (S >= 2) x = __builtin_fma((double)a1, 256.0, x);
    if constexpr (S >= 3) {
        double hi = (double)a2;
        if constexpr (S >= 4) hi = __builtin_fma((double)a3, 256.0, hi);
        if (a.epi2) {
            asm volatile("");   // a real (uniform) branch: hipcc would otherwise compute this and select
            hi = __builtin_fma(-__builtin_rint(hi * a.qinv), a.qd, hi);
        }
        x = __builtin_fma(hi, 65536.0, x);
    }
    const double t = __builtin_rint(x * a.qinv);
    const uint32_t r = (uint32_t)(int)__builtin_fma(-t, a.qd, x);
    const uint32_t rq = r + a.mod.q;
    return r < rq ? r : rq;
}
