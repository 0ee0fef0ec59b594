// Kernel (a): batched mod-q matrix product on the int8 matrix cores of gfx950.
//
//   out[b][i] = sum_j key[i][j] * X[b][j]  (+ addend[b][i]) (+ msg[b]*floor(q/2) on the last row)  mod q
//
// Both operands are split into signed radix-2^8 digits (d in [-128,127]); the product of two
// digit planes is an exact int8 x int8 -> int32 MFMA (v_mfma_i32_32x32x32_i8).  Digit pairs with
// the same weight 2^(8s) share one int32 accumulator; the epilogue recombines the S = LA+LB-1
// accumulators in int64 with centred constants c_s = 2^(8s) mod q and does one Barrett reduction.
//
// HBM layout ("fragment order"): a digit plane of an R x K matrix is stored as
//   [R/32 row blocks][K/32 k-steps][L digits][64 lanes][16 bytes]
// where lane l of a fragment holds row (l & 31), k = 32*kstep + 16*(l >> 5) + 0..15 -- exactly
// the 16 bytes that lane feeds to one MFMA.  A fragment is 1 KiB contiguous, so global->LDS
// staging and LDS->register reads are linear, coalesced and bank-conflict free with no padding.
//
// Work split: one workgroup = 8 waves (2 per SIMD) = 256 messages x (WN*32) key rows.  The key
// fragments of a k-tile are staged once in LDS and shared by the 8 waves; each wave owns one
// block of 32 messages and feeds its batch operand straight from registers:
//   BITS   : 16 bits of the message row are expanded to 16 int8 {0,1} in registers,
//   PLANES : pre-recoded digit fragments are loaded with one coalesced 16-byte load per lane.
// MFMA roles: A (rows of D) = messages, B (columns of D) = key rows, so a lane ends up owning
// one key row and 16 messages, and every store instruction writes two 128-byte runs.
#include <cstdlib>
#include "qf_internal.h"

#define QF_NW 8          // waves per workgroup
#define QF_KS 4          // k-steps (of 32) per k-tile -> BK = 128

// ---------------------------------------------------------------------------------------------
// recode: u32/i32 matrix -> signed-digit planes in fragment order
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t qf_maxpos(int L) {
    // largest value representable with L balanced digits: 127 * (256^L - 1) / 255
    int64_t p = 0;
    for (int i = 0; i < L; i++) p = p * 256 + 127;
    return p;
}

template <int L>
__global__ __launch_bounds__(256) void qf_recode_kernel(const uint8_t *__restrict__ src, int kind,
                                                        size_t stride_bytes, uint32_t rows,
                                                        uint32_t cols, QfMod mod, uint32_t rb,
                                                        uint32_t ks_pad, int8_t *__restrict__ planes,
                                                        uint32_t *__restrict__ flag) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = gid & 63;
    const uint64_t frag = gid >> 6;  // (rbi, ksi)
    if (frag >= (uint64_t)rb * ks_pad) return;
    const uint32_t ksi = frag % ks_pad, rbi = frag / ks_pad;
    const uint32_t row = rbi * 32 + (lane & 31);
    const uint32_t k0 = ksi * 32 + 16 * (lane >> 5);
    const int64_t maxpos = qf_maxpos(L);
    uint32_t dig[L][4];
#pragma unroll
    for (int l = 0; l < L; l++)
#pragma unroll
        for (int w = 0; w < 4; w++) dig[l][w] = 0;
    bool bad = false;
    if (row < rows) {
        const uint8_t *rp = src + (size_t)row * stride_bytes;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t k = k0 + j;
            int64_t rep = 0;
            if (k < cols) {
                uint32_t c;
                if (kind == QF_X_U32) {
                    c = qf_red_u64(((const uint32_t *)rp)[k], mod);
                } else {
                    int64_t v = ((const int32_t *)rp)[k];
                    c = qf_red_i64(v, mod);
                }
                rep = (int64_t)c <= maxpos ? (int64_t)c : (int64_t)c - (int64_t)mod.q;
            }
#pragma unroll
            for (int l = 0; l < L; l++) {
                int64_t d = ((rep + 128) & 255) - 128;
                dig[l][j >> 2] |= (uint32_t)(d & 255) << (8 * (j & 3));
                rep = (rep - d) >> 8;
            }
            bad |= rep != 0;
        }
    }
    int8_t *dst = planes + ((frag * L) * 64 + lane) * 16;
#pragma unroll
    for (int l = 0; l < L; l++)
        *(uint4 *)(dst + (size_t)l * 1024) = make_uint4(dig[l][0], dig[l][1], dig[l][2], dig[l][3]);
    if (bad) atomicOr(flag, 1u);
}

int qf_digits_for_modulus(uint32_t q) {
    // smallest L with q <= 256^L: every residue then has a representative with L balanced digits
    int L = 1;
    uint64_t cap = 256;
    while (cap < q) { cap <<= 8; L++; }
    return L;
}

int qf_launch_recode(qf_ctx *ctx, const void *src, int kind, size_t stride_bytes, uint32_t rows,
                     uint32_t cols, const QfMod &mod, int digits, uint32_t ks_pad, int8_t *planes,
                     uint32_t *flag) {
    uint32_t rb = (rows + 31) / 32;
    uint64_t threads = (uint64_t)rb * ks_pad * 64;
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    const uint8_t *s = (const uint8_t *)src;
    switch (digits) {
        case 1: hipLaunchKernelGGL(qf_recode_kernel<1>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag); break;
        case 2: hipLaunchKernelGGL(qf_recode_kernel<2>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag); break;
        case 3: hipLaunchKernelGGL(qf_recode_kernel<3>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag); break;
        case 4: hipLaunchKernelGGL(qf_recode_kernel<4>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag); break;
        default: return qf_fail(ctx, QF_ERR_ARG, "recode: digits must be 1..4");
    }
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------
// transpose (key upload with transpose=1)
// ---------------------------------------------------------------------------------------------
__global__ void qf_transpose_kernel(const uint32_t *__restrict__ src, uint32_t rows, uint32_t cols,
                                    uint32_t *__restrict__ dst) {
    __shared__ uint32_t tile[32][33];
    uint32_t bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (uint32_t i = threadIdx.y; i < 32; i += blockDim.y) {
        uint32_t r = by + i, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[i][threadIdx.x] = src[(size_t)r * cols + c];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.y; i < 32; i += blockDim.y) {
        uint32_t c = bx + i, r = by + threadIdx.x;  // dst is cols x rows
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[threadIdx.x][i];
    }
}
int qf_launch_transpose_u32(qf_ctx *ctx, const uint32_t *src, uint32_t rows, uint32_t cols, uint32_t *dst) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(qf_transpose_kernel, grid, block, 0, ctx->stream, src, rows, cols, dst);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------
struct QfGemmArgs {
    const int8_t *kplanes;   // key digit planes, fragment order
    uint32_t rb;             // key row blocks
    uint32_t ks;             // k-steps (multiple of QF_KS)
    uint32_t rows;           // key rows (= output row length)
    const uint8_t *xbits;    // BITS: bit-packed rows
    size_t xstride;          // BITS: row stride in bytes (multiple of 16)
    const int8_t *xplanes;   // PLANES: batch digit planes, fragment order [mb][ks][LB][64][16]
    uint64_t B;              // messages
    const uint32_t *addend;  // B x rows or null
    const uint8_t *msg;      // B or null
    uint32_t *out;           // B x rows
    QfMod mod;
    int32_t cs[8];           // centred 2^(8s) mod q
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

template <int LA, int LB, int WN, bool BITS>
__global__ __launch_bounds__(QF_NW * 64, 2) void qf_gemm_kernel(QfGemmArgs a) {
    constexpr int S = LA + LB - 1;
    // prefetch operand fragments one k-step ahead when both register sets fit next to the
    // accumulators (2 waves/SIMD => 256 VGPRs per lane)
    constexpr bool PF = WN * S * 16 + 8 * (WN * LA + LB) <= 224;
    constexpr int FRAGS = WN * QF_KS * LA;           // key fragments (1 KiB each) per k-tile
    constexpr int TILE_BYTES = FRAGS * 1024;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 * TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const uint32_t rb0 = blockIdx.y * WN;
    const uint64_t mb = (uint64_t)blockIdx.x * QF_NW + wave;  // this wave's message block
    const uint32_t ntiles = a.ks / QF_KS;

    v16i acc[WN][S];
#pragma unroll
    for (int j = 0; j < WN; j++)
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][s][r] = 0;

    // ---- key staging: the k-tile of row block (rb0+j) is QF_KS*LA consecutive 1-KiB fragments in
    // HBM; each wave copies whole fragments straight into LDS with LDS-DMA (no staging registers):
    // the destination of one wave-instruction is fragment base + lane*16, which is exactly the
    // linear fragment image the MFMA operand reads expect.
    constexpr int NSR = (FRAGS + QF_NW - 1) / QF_NW;
#define QF_STAGE(kt_, buf_)                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NSR; i_++) {                                             \
        const int f_ = wave + i_ * QF_NW;                                                            \
        if (f_ < FRAGS) {                                                                            \
            const int j_ = f_ / (QF_KS * LA), rem_ = f_ % (QF_KS * LA);                              \
            uint32_t rbi_ = rb0 + j_;                                                                \
            if (rbi_ >= a.rb) rbi_ = a.rb - 1; /* padded column tile: harmless duplicate */          \
            const int8_t *g_ = a.kplanes +                                                           \
                ((((size_t)rbi_ * a.ks + (size_t)(kt_)*QF_KS) * LA + rem_) * 64 + lane) * 16;        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_,     \
                (__attribute__((address_space(3))) void *)(smem + (buf_)*TILE_BYTES + f_ * 1024),    \
                16, 0, 0);                                                                           \
        }                                                                                            \
    }

    // ---- batch operand addressing
    const uint64_t mrow_l = mb * 32 + (lane & 31);                    // message row this lane feeds
    const uint64_t mrow_c = mrow_l < a.B ? mrow_l : (a.B - 1);        // clamped for loads
    const uint8_t *xrow = BITS ? a.xbits + mrow_c * a.xstride : nullptr;
    const uint64_t nmb = (a.B + 31) / 32;
    const uint64_t mb_c = mb < nmb ? mb : nmb - 1;
    const int8_t *xfrag = BITS ? nullptr : a.xplanes + (mb_c * a.ks * LB) * 1024 + (size_t)lane * 16;

    QF_STAGE(0, 0)
    uint4 xb = make_uint4(0, 0, 0, 0);
    if (BITS) {
        if ((size_t)16 <= a.xstride) xb = *(const uint4 *)(xrow);
    }
    __syncthreads();

    for (uint32_t kt = 0; kt < ntiles; kt++) {
        const int buf = kt & 1;
        const bool more = kt + 1 < ntiles;
        if (more) { QF_STAGE(kt + 1, buf ^ 1) }
        uint4 xb_next = make_uint4(0, 0, 0, 0);
        if (BITS && more) {
            size_t off = (size_t)(kt + 1) * 16;
            if (off + 16 <= a.xstride) xb_next = *(const uint4 *)(xrow + off);
        }
        const uint8_t *tile = smem + buf * TILE_BYTES + lane * 16;
        // Operand fragments are read one k-step ahead into a second register set so that the
        // LDS (and, for PLANES, global) latency hides behind the previous k-step's MFMAs.
        v4i bf[PF ? 2 : 1][WN * LA];
        v4i af[PF ? 2 : 1][LB];
#define QF_LOAD_B(set_, ks_)                                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                             \
        _Pragma("unroll") for (int l_ = 0; l_ < LA; l_++)                                         \
            bf[set_][j_ * LA + l_] = *(const v4i *)(tile + ((j_ * QF_KS + (ks_)) * LA + l_) * 1024);
#define QF_LOAD_A(set_, ks_)                                                                      \
    if (BITS) {                                                                                   \
        const uint32_t w_ = (ks_) == 0 ? xb.x : (ks_) == 1 ? xb.y : (ks_) == 2 ? xb.z : xb.w;     \
        af[set_][0] = qf_expand16(h ? (w_ >> 16) : (w_ & 0xFFFFu));                               \
    } else {                                                                                      \
        _Pragma("unroll") for (int b_ = 0; b_ < LB; b_++)                                         \
            af[set_][b_] = *(const v4i *)(xfrag + (((size_t)kt * QF_KS + (ks_)) * LB + b_) * 1024); \
    }
        if constexpr (PF) {
            QF_LOAD_B(0, 0)
            QF_LOAD_A(0, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < QF_KS; ks++) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < QF_KS) {
                    QF_LOAD_B(nxt, ks + 1)
                    QF_LOAD_A(nxt, ks + 1)
                }
#pragma unroll
                for (int j = 0; j < WN; j++)
#pragma unroll
                    for (int la = 0; la < LA; la++)
#pragma unroll
                        for (int b = 0; b < LB; b++)
                            acc[j][la + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[cur][b], bf[cur][j * LA + la],
                                                                                    acc[j][la + b], 0, 0, 0);
                // pin the interleave: one next-k-step LDS read issued between consecutive MFMAs of
                // this k-step (hipcc otherwise sinks every read to just before its use and waits
                // lgkmcnt(0) in front of every MFMA)
                if (ks + 1 < QF_KS) {
#pragma unroll
                    for (int i = 0; i < WN * LA; i++) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (next k-step)
                        __builtin_amdgcn_sched_group_barrier(0x008, LB, 0);  // LB MFMAs (this k-step)
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, WN * LA * LB, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // k-steps are separate scheduling regions
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < QF_KS; ks++) {
                QF_LOAD_A(0, ks)
#pragma unroll
                for (int j = 0; j < WN; j++)
#pragma unroll
                    for (int la = 0; la < LA; la++) {
                        const v4i bfr = *(const v4i *)(tile + ((j * QF_KS + ks) * LA + la) * 1024);
#pragma unroll
                        for (int b = 0; b < LB; b++)
                            acc[j][la + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0][b], bfr, acc[j][la + b], 0, 0, 0);
                    }
            }
        }
        xb = xb_next;
        __syncthreads();
    }

    // ---- epilogue: recombine digits, reduce, add noise / message, store
    const uint32_t lastrow = a.rows - 1;
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const uint32_t krow = (rb0 + j) * 32 + (lane & 31);
        if (rb0 + j >= a.rb || krow >= a.rows) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint64_t mrow = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow >= a.B) continue;
            int64_t x = 0;
#pragma unroll
            for (int s = 0; s < S; s++) x += (int64_t)acc[j][s][r] * (int64_t)a.cs[s];
            if (a.addend) x += (int64_t)a.addend[mrow * a.rows + krow];
            if (a.msg && krow == lastrow && (a.msg[mrow] & 1)) x += (int64_t)a.mod.qhalf;
            a.out[mrow * a.rows + krow] = qf_red_i64(x, a.mod);
        }
    }
}

// widest column tile (in 32-row blocks) whose accumulators fit comfortably: WN*S*16 <= 192 regs
static int qf_wn_max(int S) {
    int w = 12 / S;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <int LA, int LB, int WN, bool BITS>
static int qf_gemm_dispatch_wn(qf_ctx *ctx, const QfGemmArgs &args) {
    constexpr int FRAGS = WN * QF_KS * LA;
    size_t lds = 2 * (size_t)FRAGS * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        QF_HIP(ctx, hipFuncSetAttribute((const void *)qf_gemm_kernel<LA, LB, WN, BITS>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    uint64_t nmsgtiles = (args.B + 32 * QF_NW - 1) / (32 * QF_NW);
    dim3 grid((unsigned)nmsgtiles, (args.rb + WN - 1) / WN), block(QF_NW * 64);
    QfTimer t(ctx, &ctx->ms_gemm);
    hipLaunchKernelGGL((qf_gemm_kernel<LA, LB, WN, BITS>), grid, block, lds, ctx->stream, args);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

template <int LA, int LB, bool BITS>
static int qf_gemm_dispatch(qf_ctx *ctx, const QfGemmArgs &args) {
    constexpr int S = LA + LB - 1;
    constexpr int WMAX = (12 / S) < 1 ? 1 : ((12 / S) > 4 ? 4 : (12 / S));
    // pick the tile width that wastes the fewest padded row blocks, preferring wider tiles
    int best = 1;
    uint32_t best_waste = ~0u;
    for (int w = WMAX; w >= 1; w--) {
        uint32_t tiles = (args.rb + w - 1) / w, waste = tiles * w - args.rb;
        if (waste < best_waste) { best_waste = waste; best = w; }
    }
    if (const char *e = getenv("QF_GEMM_WN")) {  // experiments only
        int w = atoi(e);
        if (w >= 1 && w <= WMAX) best = w;
    }
    switch (best) {
        case 1: return qf_gemm_dispatch_wn<LA, LB, 1, BITS>(ctx, args);
        case 2: if constexpr (WMAX >= 2) return qf_gemm_dispatch_wn<LA, LB, 2, BITS>(ctx, args); break;
        case 3: if constexpr (WMAX >= 3) return qf_gemm_dispatch_wn<LA, LB, 3, BITS>(ctx, args); break;
        case 4: if constexpr (WMAX >= 4) return qf_gemm_dispatch_wn<LA, LB, 4, BITS>(ctx, args); break;
    }
    return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: no tile configuration");
}

int qf_launch_gemm(qf_ctx *ctx, const qf_key *key, const void *X, int x_is_bits, size_t x_stride,
                   int lb, uint64_t B, const uint32_t *addend, const uint8_t *msg, uint32_t *out) {
    (void)qf_wn_max;
    QfGemmArgs a;
    a.kplanes = key->planes; a.rb = key->rb; a.ks = key->ks; a.rows = key->rows;
    a.xbits = x_is_bits ? (const uint8_t *)X : nullptr;
    a.xstride = x_stride;
    a.xplanes = x_is_bits ? nullptr : (const int8_t *)X;
    a.B = B; a.addend = addend; a.msg = msg; a.out = out; a.mod = key->mod;
    // centred 2^(8s) mod q
    uint64_t p = 1 % key->q;
    for (int s = 0; s < 8; s++) {
        int64_t c = (int64_t)p;
        if (c > (int64_t)(key->q / 2)) c -= key->q;
        a.cs[s] = (int32_t)c;
        p = (p * 256) % key->q;
    }
    // exactness of the int64 recombination: LA*LB*K*2^14*(q/2) + 2^33 must stay below 2^62
    {
        long double bound = (long double)key->la * lb * (long double)key->ks * 32.0L * 16384.0L * (key->q / 2.0L + 1);
        if (bound > 4.0e18L) return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: K too large for exact int64 recombination");
        // int32 accumulators: min(LA,LB) pairs * K * 2^14 < 2^31
        long double ab = (long double)(key->la < lb ? key->la : lb) * key->ks * 32.0L * 16384.0L;
        if (ab >= 2147483648.0L) return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: K too large for int32 accumulation");
    }
    const int la = key->la;
#define QF_CASE(LA_, LB_)                                                            \
    if (la == LA_ && lb == LB_) {                                                    \
        return x_is_bits ? qf_gemm_dispatch<LA_, 1, true>(ctx, a)                    \
                         : qf_gemm_dispatch<LA_, LB_, false>(ctx, a);                \
    }
    if (x_is_bits && lb != 1) return qf_fail(ctx, QF_ERR_ARG, "gemm: bit operand has one digit");
    QF_CASE(1, 1) QF_CASE(2, 1) QF_CASE(3, 1) QF_CASE(4, 1)
    QF_CASE(2, 2) QF_CASE(3, 2) QF_CASE(4, 2)
    QF_CASE(3, 3) QF_CASE(4, 3)
    QF_CASE(4, 4)
#undef QF_CASE
    return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: digit combination not instantiated (need LB <= LA <= 4)");
}
