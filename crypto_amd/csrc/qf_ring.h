// Shared by qf_ntt.hip (general LDS radix-2 kernel, schoolbook fallback) and qf_ntt_wave.hip
// (wave-per-polynomial register-blocked kernel): launch arguments of the fused ring kernels.
#pragma once
#include "qf_internal.h"

enum { MODE_POLYMUL = 0, MODE_RLPR_ENC = 1, MODE_RLPR_DEC = 2, MODE_FDH = 3, MODE_FWD_ONLY = 4 };

struct QfRingArgs {
    uint32_t n, lg;
    QfMod mod;
    const uint32_t *tw;       // [4][n]
    uint32_t ninv, ninv_p;
    uint64_t B;
    // operands (meaning depends on MODE)
    const uint32_t *in0, *in1, *in2;  // POLYMUL: a,b | ENC: r,e1,e2 | DEC: u,v | FDH: sigma(i32),-,-
    const uint8_t *mu;                // ENC: message bits
    const uint32_t *key, *key_p;      // NTT-domain key polys + Shoup companions (npoly x n)
    const uint32_t *key_coef;         // coefficient-domain key polys (schoolbook path)
    uint32_t npoly;
    uint32_t *out0, *out1;            // POLYMUL: c | ENC: u,v | FDH: t | FWD_ONLY: hat, hat'
    uint8_t *outb;                    // DEC: packed bits
    // wave kernel (qf_ntt_wave.hip) only
    uint32_t one_p;                   // floor(2^32 / q): Shoup companion of 1 (cheap 32-bit reduction)
    uint32_t off32;                   // smallest multiple of q that is >= 2^31 (signed -> unsigned shift)
    // Montgomery pointwise product of the wave kernel's POLYMUL mode (q odd): q^-1 mod 2^32, and
    // n^-1 * 2^32 mod q with its Shoup companion (the final scaling absorbs the 2^-32 of the product)
    uint32_t qinv32, ninv_m, ninv_m_p;
    const uint32_t *u;                // FDH: hashed targets (B x n)
    uint8_t *ok;                      // FDH: verdicts
    uint64_t norm_bound;              // FDH: floor(s^2 * n * mbar)
    unsigned long long *dbg = nullptr; // diagnostic build only (QF_STAMP): per-wave cycle sums of the POLYMUL path
};

// qf_ntt_wave.hip: wave-per-polynomial kernels for n = 256..2048 on NTT-friendly q.  Ring keys of
// such rings keep (value, Shoup companion) pairs interleaved in `hat` (2 * npoly * n words), in
// the kernel's own register order.
bool qf_ring_wave_supported(uint32_t lg);
int qf_launch_ring_wave(qf_ctx *ctx, const qf_ring *ring, int mode, const QfRingArgs &a);

