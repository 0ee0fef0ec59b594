// Launch arguments and helpers of the mod-q GEMM (qf_gemm.hip).
#pragma once
#include "qf_internal.h"

struct QfGemmArgs {
    const int8_t *kplanes;   // key digit planes, fragment order
    uint32_t rb;             // key row blocks
    uint32_t ks;             // k-steps (multiple of QF_KS)
    uint32_t rows;           // key rows (= output row length)
    const uint8_t *xbits;    // BITS: bit-packed rows
    size_t xstride;          // BITS: row stride in bytes (multiple of 16)
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


