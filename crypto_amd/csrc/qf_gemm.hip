// Kernel (a): batched mod-q matrix product on the int8 matrix cores of gfx950.
//
//   out[b][i] = sum_j key[i][j] * X[b][j]  (+ addend[b][i]) (+ msg[b]*floor(q/2) on the last row)  mod q
//
// Both operands are split into signed radix-2^8 digits (d in [-128,127]); the product of two
// digit planes is an exact int8 x int8 -> int32 MFMA (v_mfma_i32_16x16x64_i8 for bit operands,
// v_mfma_i32_32x32x32_i8 for digit-plane operands).  Digit pairs with the same weight 2^(8s) share
// one int32 accumulator; the epilogue recombines the S = LA+LB-1 accumulators and reduces mod q --
// in fp64 for one-digit operands (every default path: qf_epi_f64 in qf_gemm.h), in int64 with
// centred constants c_s = 2^(8s) mod q and one Barrett reduction for the fused <LA,LB> kernels.
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
// one key row and 16 messages, and every store instruction writes two 128-byte runs (32x32x32;
// four 64-byte runs with 16x16x64).  The other way round (a lane owns one message and 4 consecutive
// key rows, 16-byte stores) was measured: 12 % slower on LPR -- a store that touches 32 rows costs
// the memory pipeline more than the instructions it saves (profiles/r2/gemm_ablation.log).
#include <cstdlib>
#include "qf_internal.h"

#define QF_NW 8          // waves per workgroup
#ifndef QF_PL_D4
#define QF_PL_D4 0
#endif
#ifndef QF_EPI_AUX_LD
#define QF_EPI_AUX_LD 0  // cache policy of the epilogue's addend loads (2 = nt)
#endif
#ifndef QF_EPI_AUX_ST
#define QF_EPI_AUX_ST 0  // cache policy of the epilogue's output stores (2 = nt)
#endif
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
                                                        uint32_t *__restrict__ flag, uint32_t mult,
                                                        uint32_t fold, uint32_t fold_b) {
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
        // the lane's 16 consecutive entries (64 bytes): four 16-byte loads when the run is aligned
        // and inside the row, single words otherwise
        uint32_t raw[16];
        if (k0 + 16 <= cols && ((((uintptr_t)rp) + (size_t)k0 * 4) & 15) == 0) {
            const uint4 *vp = (const uint4 *)(rp + (size_t)k0 * 4);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const uint4 x = vp[v];
                raw[4 * v] = x.x; raw[4 * v + 1] = x.y; raw[4 * v + 2] = x.z; raw[4 * v + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) raw[j] = k0 + j < cols ? ((const uint32_t *)rp)[k0 + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t k = k0 + j;
            int64_t rep = 0;
            if (k < cols) {
                uint32_t c;
                if (kind == QF_X_U32) {
                    c = qf_red_u64(raw[j], mod);
                } else {
                    int64_t v = (int32_t)raw[j];
                    c = qf_red_i64(v, mod);
                }
                if (mult != 1) c = qf_red_u64((uint64_t)c * mult, mod);   // folded key planes: 256^b * A mod q
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
    int8_t *dst = planes + (((frag * fold + fold_b) * L) * 64 + lane) * 16;
#pragma unroll
    for (int l = 0; l < L; l++)
        *(uint4 *)(dst + (size_t)l * 1024) = make_uint4(dig[l][0], dig[l][1], dig[l][2], dig[l][3]);
    if (bad) atomicOr(flag, 1u);
}

// packed remainder block of a key (see qf_gemm_body PACKED): one fragment per k-step whose row
// i*L + l is digit l of trailing row `first_row + i`
template <int L>
__global__ __launch_bounds__(256) void qf_recode_rem_kernel(const uint32_t *__restrict__ src, uint32_t cols,
                                                            uint32_t first_row, uint32_t rem_rows, QfMod mod,
                                                            uint32_t ks_pad, int8_t *__restrict__ planes) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = gid & 63;
    const uint64_t ksi = gid >> 6;
    if (ksi >= ks_pad) return;
    const uint32_t p = lane & 31, ir = p / L, l = p % L;
    const uint32_t k0 = (uint32_t)ksi * 32 + 16 * (lane >> 5);
    const int64_t maxpos = qf_maxpos(L);
    uint32_t dig[4] = {0, 0, 0, 0};
    if (ir < rem_rows) {
        const uint32_t *rp = src + (size_t)(first_row + ir) * cols;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t k = k0 + j;
            int64_t rep = 0;
            if (k < cols) {
                const uint32_t c = qf_red_u64(rp[k], mod);
                rep = (int64_t)c <= maxpos ? (int64_t)c : (int64_t)c - (int64_t)mod.q;
            }
            int64_t d = 0;
            for (uint32_t t = 0; t <= l; t++) { d = ((rep + 128) & 255) - 128; rep = (rep - d) >> 8; }
            dig[j >> 2] |= (uint32_t)(d & 255) << (8 * (j & 3));
        }
    }
    *(uint4 *)(planes + (ksi * 64 + lane) * 16) = make_uint4(dig[0], dig[1], dig[2], dig[3]);
}
int qf_launch_recode_rem(qf_ctx *ctx, const uint32_t *src, uint32_t cols, uint32_t first_row, uint32_t rem_rows,
                         const QfMod &mod, int digits, uint32_t ks_pad, int8_t *planes) {
    dim3 grid((unsigned)(((uint64_t)ks_pad * 64 + 255) / 256)), block(256);
    switch (digits) {
        case 1: hipLaunchKernelGGL(qf_recode_rem_kernel<1>, grid, block, 0, ctx->stream, src, cols, first_row, rem_rows, mod, ks_pad, planes); break;
        case 2: hipLaunchKernelGGL(qf_recode_rem_kernel<2>, grid, block, 0, ctx->stream, src, cols, first_row, rem_rows, mod, ks_pad, planes); break;
        case 3: hipLaunchKernelGGL(qf_recode_rem_kernel<3>, grid, block, 0, ctx->stream, src, cols, first_row, rem_rows, mod, ks_pad, planes); break;
        case 4: hipLaunchKernelGGL(qf_recode_rem_kernel<4>, grid, block, 0, ctx->stream, src, cols, first_row, rem_rows, mod, ks_pad, planes); break;
        default: return qf_fail(ctx, QF_ERR_ARG, "recode: digits must be 1..4");
    }
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
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
                     uint32_t *flag, uint32_t mult, uint32_t fold, uint32_t fold_b) {
    uint32_t rb = (rows + 31) / 32;
    uint64_t threads = (uint64_t)rb * ks_pad * 64;
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    const uint8_t *s = (const uint8_t *)src;
    switch (digits) {
        case 1: hipLaunchKernelGGL(qf_recode_kernel<1>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag, mult, fold, fold_b); break;
        case 2: hipLaunchKernelGGL(qf_recode_kernel<2>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag, mult, fold, fold_b); break;
        case 3: hipLaunchKernelGGL(qf_recode_kernel<3>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag, mult, fold, fold_b); break;
        case 4: hipLaunchKernelGGL(qf_recode_kernel<4>, grid, block, 0, ctx->stream, s, kind, stride_bytes, rows, cols, mod, rb, ks_pad, planes, flag, mult, fold, fold_b); break;
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
#include <mutex>
#include "qf_gemm.h"

#ifdef QF_STAMP
#define QF_T(var_) unsigned long long var_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var_) :: "memory"); __builtin_amdgcn_sched_barrier(0);
#define QF_ACC(sum_, a_, b_) sum_ += (b_) - (a_);
#else
#define QF_T(var_)
#define QF_ACC(sum_, a_, b_)
#endif
static unsigned long long *g_qf_dbg = nullptr;
extern "C" void qf_debug_set_buffer(void *p) { g_qf_dbg = (unsigned long long *)p; }

// Pins the instruction interleave of one k-step for hipcc's scheduler: N groups of (1 LDS read,
// MF MFMAs), with NV vector-memory issues (the LDS-DMA pieces) spread evenly between them.
template <int I, int N, int NV, int MF>
struct QfSched {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
        constexpr int nv = (I + 1) * NV / N - I * NV / N;
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x010, nv, 0);
        if constexpr (I + 1 < N) QfSched<I + 1, N, NV, MF>::run();
    }
};

// Group of k-steps 2..3 (groups 2*NBF .. 4*NBF-1 of a tile) behind which LDS-DMA piece p of NSR is
// issued: evenly spread, the last piece behind the tile's last group.
__host__ __device__ constexpr int qf_piece_group(int p, int nsr, int nbf) { return 2 * nbf + ((p + 1) * 2 * nbf) / nsr - 1; }
// pieces issued before k-step 3 starts (the counted wait for the tile's batch-operand load skips them)
__host__ __device__ constexpr int qf_pieces_before_ks3(int nsr, int nbf) {
    int c = 0;
    for (int p = 0; p < nsr; p++) c += qf_piece_group(p, nsr, nbf) < 3 * nbf ? 1 : 0;
    return c;
}

// 16x16x64 body: a tile is G = 2H groups; the NSR LDS-DMA pieces go behind groups of its second half
__host__ __device__ constexpr int qf16_piece_group(int p, int nsr, int h) { return h + ((p + 1) * h) / nsr - 1; }
__host__ __device__ constexpr int qf16_pieces_before(int nsr, int h, int g) {
    int c = 0;
    for (int p = 0; p < nsr; p++) c += qf16_piece_group(p, nsr, h) < g ? 1 : 0;
    return c;
}

// PACKED (bit operands, LB = 1 only): the tile's last row block is the key's "remainder block" --
// the r = rows mod 32 trailing rows, whose r*LA digit rows are packed into ONE fragment per k-step
// (fragment row i*LA + l = digit l of trailing row i; a.rem_planes) instead of LA mostly-empty ones.
// With LB = 1 every digit has its own accumulator anyway, so digits may sit in different fragment
// rows; the epilogue gathers the LA partial sums of a row from neighbouring lanes.  Regev n = 256:
// 257 rows cost 8*3 + 1 = 25 MFMAs per k-step and message block instead of 27.
template <int LA, int LB, int WN, int MW, bool BITS, bool PACKED>
__device__ __forceinline__ void qf_gemm_body(const QfGemmArgs &a, const uint32_t bx, const uint32_t rb_first) {
    static_assert(!PACKED || (BITS && LB == 1 && MW == 1), "packed remainder block: bit operands only");
    constexpr int S = LA + LB - 1;
    constexpr int NBF = PACKED ? (WN - 1) * LA + 1 : WN * LA;   // key fragments per k-step
#define QF_LJ(j_) ((PACKED && (j_) == WN - 1) ? 1 : LA)         /* digits (fragments) of row block j */
#define QF_FOFF(j_, ks_, l_) ((PACKED && (j_) == WN - 1) ? ((WN - 1) * QF_KS * LA + (ks_)) : (((j_) * QF_KS + (ks_)) * LA + (l_)))
    // prefetch operand fragments one k-step ahead when both register sets fit next to the
    // accumulators (2 waves/SIMD => 256 VGPRs per lane)
    constexpr bool PF = MW * WN * S * 16 + 8 * (WN * LA + MW * LB) <= 224;
    constexpr int FRAGS = NBF * QF_KS;               // key fragments (1 KiB each) per k-tile
    constexpr int TILE_BYTES = FRAGS * 1024;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 3 * TILE_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const uint32_t rb0 = rb_first;   // first row block of this column tile
    const uint64_t mb0 = ((uint64_t)bx * QF_NW + wave) * MW;  // this wave's first message block
    const uint32_t ntiles = a.ks / QF_KS;

    v16i acc[MW][WN][S];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][s][r] = 0;

    // ---- key staging: the k-tile of row block (rb0+j) is QF_KS*LA consecutive 1-KiB fragments in
    // HBM; each wave copies whole fragments straight into LDS with LDS-DMA (no staging registers):
    // the destination of one wave-instruction is fragment base + lane*16, which is exactly the
    // linear fragment image the MFMA operand reads expect.
    constexpr int NSR = (FRAGS + QF_NW - 1) / QF_NW;
    // Every wave issues exactly NSR DMA instructions per tile (a wave whose share runs past the
    // tile re-copies the last fragment: same bytes to the same place), so the stream has no
    // branches and hipcc can count vmcnt exactly around it.
    // A piece's source is (tile-0 pointer of its fragment, loop-invariant, scalar) + tile * (bytes per
    // k-tile of its row block) + lane * 16: two scalar adds per piece in the loop (hipcc re-derived the
    // whole 64-bit product chain, 9 scalar instructions, for every piece of every tile).
#define QF_STAGE(kt_, buf_) QF_STAGE_PART(kt_, buf_, 0, NSR)
#define QF_STAGE_SETUP()                                                                             \
    const int8_t *stg_ptr[NSR];                                                                      \
    uint32_t stg_stride[PACKED ? NSR : 1];                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < NSR; i_++) {                                             \
        int f_ = wave + i_ * QF_NW;                                                                  \
        f_ = f_ < FRAGS ? f_ : FRAGS - 1;                                                            \
        const int j_ = f_ / (QF_KS * LA), rem_ = f_ % (QF_KS * LA);                                  \
        uint32_t rbi_ = rb0 + j_;                                                                    \
        rbi_ = rbi_ < a.rb ? rbi_ : a.rb - 1; /* padded column tile: harmless duplicate */           \
        stg_ptr[i_] = a.kplanes + ((size_t)rbi_ * a.ks * LA + rem_) * 1024;                          \
        if constexpr (PACKED) {                                                                      \
            stg_stride[i_] = QF_KS * LA * 1024;                                                      \
            if (f_ >= (WN - 1) * QF_KS * LA) {                                                       \
                stg_ptr[i_] = a.rem_planes + (size_t)(f_ - (WN - 1) * QF_KS * LA) * 1024;            \
                stg_stride[i_] = QF_KS * 1024;                                                       \
            }                                                                                        \
        }                                                                                            \
        asm volatile("" : "+s"(stg_ptr[i_]));                                                        \
    }                                                                                                \
    const uint32_t stg_lane16 = (uint32_t)lane * 16;
#define QF_STAGE_PART(kt_, buf_, i0_, i1_)                                                           \
    _Pragma("unroll") for (int i_ = (i0_); i_ < (i1_); i_++) {                                       \
        const int8_t *t_ = stg_ptr[i_] + (size_t)(kt_) * (PACKED ? stg_stride[i_] : (uint32_t)(QF_KS * LA * 1024)); \
        asm volatile("" : "+s"(t_));   /* scalar: keeps hipcc from hoisting NSR per-lane 64-bit addresses out of the loop */ \
        const int8_t *g_ = t_ + stg_lane16;                                                          \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_,         \
            (__attribute__((address_space(3))) void *)(smem + (buf_)*TILE_BYTES + (wave + i_ * QF_NW < FRAGS ? wave + i_ * QF_NW : FRAGS - 1) * 1024), \
            16, 0, 0);                                                                               \
    }

    // ---- batch operand addressing (MW message blocks of 32 per wave)
    const uint64_t nmb = (a.B + 31) / 32;
    const uint8_t *xrow[MW];
    const int8_t *xfrag[MW];
#pragma unroll
    for (int i = 0; i < MW; i++) {
        const uint64_t mrow_l = (mb0 + i) * 32 + (lane & 31);             // message row this lane feeds
        const uint64_t mrow_c = mrow_l < a.B ? mrow_l : (a.B - 1);        // clamped for loads
        xrow[i] = BITS ? a.xbits + mrow_c * a.xstride : nullptr;
        const uint64_t mb_c = mb0 + i < nmb ? mb0 + i : nmb - 1;
        xfrag[i] = BITS ? nullptr : a.xplanes + ((mb_c * a.ks * a.xlb_total) + a.xlb_off) * 1024 + (size_t)lane * 16;
    }

    // ---- main loop.  Three LDS buffers: the DMA of k-tile t+2 is issued in the middle of tile t
    // (about 1.5 tiles before its first use), so its ~1 us issue-to-landed latency never shows.
    // One barrier per tile, placed mid-tile: every wave that passes it has (a) finished reading
    // tile t-1's buffer, which the new DMA overwrites, and (b) seen its own DMA of tile t+1 land
    // (vmcnt(0) in front of the barrier), so the last k-step of tile t can already prefetch the
    // first fragments of tile t+1 and the MFMA stream runs across tile boundaries without a bubble.
    // The batch operand of tile t+1 (16 bytes of bits, or KS*LB fragments) is loaded to registers
    // at the same point but BEFORE the DMA is issued: vmcnt retires in order, so a load issued
    // after the DMA could not be consumed until the DMA had landed.
    constexpr int NAT = BITS ? 1 : QF_KS * LB;        // register sets (16 B) of one tile's batch operand
    constexpr bool ATILE = BITS || MW * NAT <= 8;     // whole-tile register prefetch affordable
    v4i at[2][MW][ATILE ? NAT : 1];
    v4i bf[PF ? 2 : 1][WN * LA];
    v4i af[PF ? 2 : 1][MW][LB];
    // The tile's batch-operand loads are inline asm: hipcc waits vmcnt(0) for any compiler-visible
    // load while an LDS-DMA is in flight, which would stall on the DMA issued right behind them.
    // QF_WAIT_AT is the matching counted wait (all but the NSR younger DMA instructions retired).
#define QF_LOAD_AT(set_, kt_)                                                                     \
    _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++) {                                           \
    if (BITS) {                                                                                   \
        /* bits past the row's stride only ever meet zero-padded key columns: clamp, no branch */ \
        size_t off_ = (size_t)(kt_)*16;                                                           \
        off_ = off_ + 16 <= a.xstride ? off_ : a.xstride - 16;                                    \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(at[set_][m_][0]) : "v"(xrow[m_] + off_) : "memory"); \
    } else if (ATILE) {                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NAT; i_++)                                        \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(at[set_][m_][i_])               \
                         : "v"(xfrag[m_] + (((size_t)(kt_)*QF_KS + i_ / LB) * a.xlb_total + i_ % LB) * 1024) : "memory"); \
    } }
#define QF_WAIT_AT(set_, n_)                                                                      \
    if (BITS || ATILE) {                                                                          \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory");                                 \
        _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++)                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < NAT; i_++) asm volatile("" : "+v"(at[set_][m_][i_])); \
    }
#ifdef QF_ABL_NOLDS
#define QF_LOAD_B(set_, tile_, ks_)                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                             \
        _Pragma("unroll") for (int l_ = 0; l_ < QF_LJ(j_); l_++)                                  \
            asm volatile("" : "+v"(bf[set_][j_ * LA + l_]));
#else
#define QF_LOAD_B(set_, tile_, ks_)                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                             \
        _Pragma("unroll") for (int l_ = 0; l_ < QF_LJ(j_); l_++)                                  \
            bf[set_][j_ * LA + l_] = *(const v4i *)((tile_) + QF_FOFF(j_, ks_, l_) * 1024);
#endif
#define QF_LOAD_A(set_, atset_, kt_, ks_)                                                         \
    _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++) {                                           \
    if (BITS) {                                                                                   \
        const uint32_t w_ = (uint32_t)at[atset_][m_][0][ks_];                                     \
        af[set_][m_][0] = qf_expand16(h ? (w_ >> 16) : (w_ & 0xFFFFu));                           \
    } else if (ATILE) {                                                                           \
        _Pragma("unroll") for (int b_ = 0; b_ < LB; b_++) af[set_][m_][b_] = at[atset_][m_][(ks_)*LB + b_]; \
    } else {                                                                                      \
        _Pragma("unroll") for (int b_ = 0; b_ < LB; b_++)                                         \
            af[set_][m_][b_] = *(const v4i *)(xfrag[m_] + (((size_t)(kt_)*QF_KS + (ks_)) * a.xlb_total + b_) * 1024); \
    } }
#define QF_MFMAS(set_)                                                                            \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                             \
        _Pragma("unroll") for (int l_ = 0; l_ < QF_LJ(j_); l_++)                                  \
            _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++)                                     \
                _Pragma("unroll") for (int b_ = 0; b_ < LB; b_++)                                 \
                    acc[m_][j_][l_ + b_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[set_][m_][b_], bf[set_][j_ * LA + l_], \
                                                                                 acc[m_][j_][l_ + b_], 0, 0, 0);
    // pin the interleave: one prefetch LDS read between consecutive MFMA groups (hipcc otherwise
    // sinks every read to just before its use and waits lgkmcnt(0) in front of every MFMA)
#define QF_INTERLEAVE()                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < NBF; i_++) {                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, MW * LB, 0);                                  \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);

    // same, with NV_ LDS-DMA issues spread evenly between the MFMA groups: a DMA issue holds its wave
    // for ~60-180 cycles, and a burst of them right behind the barrier stalls both waves of a SIMD
    // at once; one at a time, the partner wave's MFMAs cover it
#define QF_INTERLEAVE_V(NV_) QfSched<0, NBF, (NV_), MW * LB>::run(); __builtin_amdgcn_sched_barrier(0);

    QF_LOAD_AT(0, 0)
    QF_STAGE_SETUP()
    QF_STAGE(0, 0)
    QF_STAGE(ntiles > 1 ? 1 : 0, 1)
    // message bits of this wave's MW*32 messages as one 32-bit mask per block (SGPRs), loaded here so
    // that the wait merges with the prologue's drain; only tiles that hold the key's last row need it
    // (the epilogue used to load one byte per element there, each load waited for on its own)
    uint32_t msgmask[MW];
#pragma unroll
    for (int i = 0; i < MW; i++) msgmask[i] = 0;
    if (a.msg && (uint64_t)(rb0 + WN) * 32 >= a.rows) {
#pragma unroll
        for (int i = 0; i < MW; i++) {
            const uint64_t mrow_l = (mb0 + i) * 32 + (lane & 31);
            const bool bit = lane < 32 && mrow_l < a.B && (a.msg[mrow_l < a.B ? mrow_l : 0] & 1);
            msgmask[i] = (uint32_t)__ballot(bit);
        }
    }
    QF_WAIT_AT(0, 0)
    __syncthreads();  // drains the DMA (vmcnt(0)) and synchronises the workgroup
#ifdef QF_STAMP
    unsigned long long s_vm = 0, s_bar = 0, s_at = 0, s_loop = 0, s_epi = 0, s_eld = 0, s_ecomp = 0, s_edrain = 0;
#endif
    // ---- main loop, pinned form (bit operands and register-resident digit operands).  The key
    // fragments of a tile are ONE stream of NF = NBF*KS reads that runs D fragments ahead of the
    // MFMAs through a ring of D registers, across k-steps and across tiles.  hipcc cannot be
    // trusted with it: it orders LDS reads against the LDS-DMA issues as may-alias stores (so every
    // read that program order puts before a DMA piece bunches up in front of it), and its counted
    // waits collapse to lgkmcnt(0) behind the loop's back edge and behind every such bunch -- four
    // full LDS round trips per tile on both waves of a SIMD at once.  Here the reads and their
    // counted waits are inline asm (the compiler sees neither a load nor a wait to second-guess),
    // every group -- wait for fragment g, its MW*LB MFMAs, the read of fragment g+D into the slot
    // just consumed, at most one DMA piece -- ends in a scheduling barrier, and
    // check_asm_window.py proves on the ISA that nothing touches a ring register between its read
    // and the wait that covers it.
    constexpr bool PINNED = BITS || ATILE;
#ifdef QF_GEMM_OLD_LOOP
    constexpr bool USE_PINNED = false;
#else
    constexpr bool USE_PINNED = PINNED && MW == 1;   // MW = 2 (experiments only) has no registers for the ring
#endif
    // fragments in flight per wave (16 registers).  A read may reach into tile t+1 only behind tile
    // t's mid-tile barrier (which is what guarantees that tile t+1 has landed): D <= 2*NBF.
    constexpr int ACCREGS = PACKED ? ((WN - 1) * S + 1) * 16 * MW : MW * WN * S * 16;
    constexpr int D = PF ? NBF : ((NBF >= 2 && ACCREGS <= 192) ? 4 : 2);   // a whole k-step ahead where the registers allow
    constexpr int NF = NBF * QF_KS;
    static_assert(NF % D == 0 && D <= 2 * NBF, "ring slots must repeat per tile; reads must not pass the mid-tile barrier");
    QF_T(tl0)
    uint32_t buf = 0;
    if constexpr (USE_PINNED) {
#define QF_FRAG_OFF(fi_, ks_) ((PACKED && (fi_) == NBF - 1) ? ((WN - 1) * QF_KS * LA + (ks_)) : ((((fi_) / LA) * QF_KS + (ks_)) * LA + (fi_) % LA))
#define QF_RD(slot_, base_, fi_, ks_)                                                             \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[slot_]) : "v"(base_), "n"(QF_FRAG_OFF(fi_, ks_) * 1024));
        v4i ring[D];
        v4i afb[2][MW];
        uint32_t wsel[MW];
        const uint32_t lds_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem + (uint32_t)lane * 16;
#pragma unroll
        for (int g = 0; g < D; g++) QF_RD(g, lds_lane, g % NBF, g / NBF)
        if constexpr (BITS) {
#pragma unroll
            for (int m = 0; m < MW; m++) {
                const uint32_t w = (uint32_t)at[0][m][0][0];
                afb[0][m] = qf_expand16(h ? (w >> 16) : (w & 0xFFFFu));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int P2 = qf_pieces_before_ks3(NSR, NBF);
        for (uint32_t kt = 0; kt < ntiles; kt++) {
            const uint32_t buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
            const uint32_t tile = lds_lane + buf * TILE_BYTES, tile1 = lds_lane + buf1 * TILE_BYTES;
            const uint32_t kt1 = kt + 1 < ntiles ? kt + 1 : kt;  // clamped: the tail prefetches are discarded
            const uint32_t kt2 = kt + 2 < ntiles ? kt + 2 : ntiles - 1;
#pragma unroll
            for (int ks = 0; ks < QF_KS; ks++) {
#ifndef QF_ABL_NOPRIO
                // The two waves of a SIMD take turns at the higher issue priority, half a barrier interval
                // each.  Left alone the arbiter favours the older wave (0-3) all the time: it finishes its
                // k-steps early, idles at the barrier for a third of the loop (per-wave stamps: 20.4 k of
                // 60.6 k cycles, waves 4-7: 1.7 k) and leaves its partner alone on the pipe, where one wave
                // cannot hide its own LDS latency.
                if (ks == 0) { if (wave >= QF_NW / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
                if (ks == 2) { if (wave >= QF_NW / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
                if (ks == 2) {
                    // mid-tile: tile t+1 has landed for everyone, tile t-1's buffer is free
                    QF_T(ta)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    QF_T(tb)
#ifndef QF_ABL_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                    QF_T(tc)
                    QF_ACC(s_vm, ta, tb) QF_ACC(s_bar, tb, tc)
                    QF_LOAD_AT(1, kt1)
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ks == 3) {
                    QF_WAIT_AT(1, P2)
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int fi = 0; fi < NBF; fi++) {
                    const int g = ks * NBF + fi, slot = g % D;
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(D - 1));
                    asm volatile("" : "+v"(ring[slot]));
                    const int j = (PACKED && fi == NBF - 1) ? WN - 1 : fi / LA;
                    const int l = (PACKED && fi == NBF - 1) ? 0 : fi % LA;
#pragma unroll
                    for (int m = 0; m < MW; m++)
#pragma unroll
                        for (int b = 0; b < LB; b++) {
                            if constexpr (BITS)
                                acc[m][j][l + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afb[ks & 1][m], ring[slot], acc[m][j][l + b], 0, 0, 0);
                            else
                                acc[m][j][l + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(at[0][m][ks * LB + b], ring[slot], acc[m][j][l + b], 0, 0, 0);
                        }
                    const int gn = g + D;
#ifndef QF_ABL_NOLDS
                    if (gn < NF) { QF_RD(slot, tile, gn % NBF, gn / NBF) }
                    else { QF_RD(slot, tile1, (gn - NF) % NBF, (gn - NF) / NBF) }
#endif
#ifndef QF_ABL_NOEXP
                    if constexpr (BITS) {
                        // the batch operand of the NEXT k-step: 16 bits -> 16 bytes in registers, one
                        // dword (4 VALU instructions) per group so that it never delays an MFMA
                        if (fi == 0) {
#pragma unroll
                            for (int m = 0; m < MW; m++) {
                                const uint32_t w = ks < 3 ? (uint32_t)at[0][m][0][(ks + 1) & 3] : (uint32_t)at[1][m][0][0];
                                wsel[m] = h ? (w >> 16) : (w & 0xFFFFu);
                            }
                        }
#pragma unroll
                        for (int d = 0; d < 4; d++)
                            if (NBF >= 4 ? fi == d : fi == 0) {
#pragma unroll
                                for (int m = 0; m < MW; m++)
                                    afb[(ks + 1) & 1][m][d] = (int)((((wsel[m] >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
                            }
                    }
#else
                    if (BITS && fi == 0 && ks == 0) { _Pragma("unroll") for (int m = 0; m < MW; m++) afb[1][m] = afb[0][m]; }
#endif
#ifndef QF_ABL_NODMA
                    if (ks >= 2) {
#pragma unroll
                        for (int p = 0; p < NSR; p++)
                            if (qf_piece_group(p, NSR, NBF) == g) { QF_STAGE_PART(kt2, buf2, p, p + 1) }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int m = 0; m < MW; m++)
#pragma unroll
                for (int i = 0; i < NAT; i++) at[0][m][i] = at[1][m][i];
            buf = buf1;
        }
        // the reads issued for a tile past the end must not land in registers the epilogue re-uses
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int g = 0; g < D; g++) asm volatile("" : "+v"(ring[g]));
    } else {
    if constexpr (PF) {
        QF_LOAD_B(0, smem + lane * 16, 0)
        QF_LOAD_A(0, 0, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t kt = 0; kt < ntiles; kt++) {
        const uint32_t buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        const uint8_t *tile = smem + buf * TILE_BYTES + lane * 16;
        const uint8_t *tile1 = smem + buf1 * TILE_BYTES + lane * 16;
        const uint32_t kt1 = kt + 1 < ntiles ? kt + 1 : kt;  // clamped: the tail prefetches are discarded
        const uint32_t kt2 = kt + 2 < ntiles ? kt + 2 : ntiles - 1;
        if constexpr (PF) {
            QF_LOAD_B(1, tile, 1) QF_LOAD_A(1, 0, kt, 1) QF_MFMAS(0) QF_INTERLEAVE()
            QF_LOAD_B(0, tile, 2) QF_LOAD_A(0, 0, kt, 2) QF_MFMAS(1) QF_INTERLEAVE()
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                QF_LOAD_A(0, 0, kt, ks) QF_LOAD_B(0, tile, ks) QF_MFMAS(0)
            }
        }
        // mid-tile: tile t+1 has landed for everyone, tile t-1's buffer is free
        QF_T(ta)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QF_T(tb)
#ifndef QF_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        QF_T(tc)
        QF_ACC(s_vm, ta, tb) QF_ACC(s_bar, tb, tc)
        QF_LOAD_AT(1, kt1)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PF) {
            // k-steps 2, 3; the last one prefetches k-step 0 of the next tile.  The DMA of tile
            // t+2 is issued piecewise between their MFMAs.
            constexpr int NSR1 = (NSR + 1) / 2;
            QF_LOAD_B(1, tile, 3) QF_LOAD_A(1, 0, kt, 3)
#ifndef QF_ABL_NODMA
            QF_STAGE_PART(kt2, buf2, 0, NSR1)
#endif
            QF_MFMAS(0) QF_INTERLEAVE_V(NSR1)
            QF_T(td)
            QF_WAIT_AT(1, NSR1)
            QF_T(te)
            QF_ACC(s_at, td, te)
            QF_LOAD_B(0, tile1, 0) QF_LOAD_A(0, 1, kt1, 0)
#ifndef QF_ABL_NODMA
            QF_STAGE_PART(kt2, buf2, NSR1, NSR)
#endif
            QF_MFMAS(1) QF_INTERLEAVE_V(NSR - NSR1)
        } else {
#ifndef QF_ABL_NODMA
            QF_STAGE(kt2, buf2)
#endif
#pragma unroll
            for (int ks = 2; ks < 4; ks++) {
                QF_LOAD_A(0, 0, kt, ks) QF_LOAD_B(0, tile, ks) QF_MFMAS(0)
            }
            QF_WAIT_AT(1, NSR)
        }
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int i = 0; i < (ATILE ? NAT : 1); i++) at[0][m][i] = at[1][m][i];
        buf = buf1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must not outlive the workgroup's LDS
    __builtin_amdgcn_s_setprio(0);
    QF_T(tl1)
    QF_ACC(s_loop, tl0, tl1)

    // ---- epilogue: recombine digits, reduce, add noise / message, store.  The lane index is re-derived
    // (mbcnt, opaque to the compiler) so that none of the per-lane offsets is computed in front of the
    // main loop and kept alive across it.
    const uint32_t lastrow = a.rows - 1;
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const uint32_t he = lane_e >> 5;
    const uint32_t rowbytes = a.rows * 4;
    const uint32_t voff_st = 4 * he * rowbytes + (lane_e & 31) * 4;                    // byte offset of (message 4h, column lane & 31)
#pragma unroll
    for (int i = 0; i < MW; i++) {
#pragma unroll
    for (int j = 0; j < WN; j++) {
        if (PACKED && j == WN - 1) {
            // remainder block: lane p = i*LA + l holds digit l of trailing row i; gather the LA partial
            // sums of a row from the neighbouring lanes, the lane with l = 0 reduces and stores
            const uint32_t p = lane & 31, ir = p / LA, l0 = p % LA;
            const uint32_t krow = (rb0 + j) * 32 + ir;
            const bool owner = l0 == 0 && ir < a.rem_rows;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint64_t mrow = (mb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                int64_t x = 0;
#pragma unroll
                for (int l = 0; l < LA; l++)
                    x += (int64_t)__shfl(acc[i][j][0][r], (int)(lane - l0 + l), 64) * (int64_t)a.cs[l];
                if (!owner || mrow >= a.B) continue;
                if (a.addend) x += (int64_t)a.addend[mrow * a.rows + krow];
                if (krow == lastrow && ((msgmask[i] >> ((r & 3) + 8 * (r >> 2) + 4 * h)) & 1)) x += (int64_t)a.mod.qhalf;
                a.out[mrow * a.rows + krow] = qf_red_i64(x, a.mod);
            }
            continue;
        }
        const uint32_t krow = (rb0 + j) * 32 + (lane_e & 31);
        if (rb0 + j >= a.rb || krow >= a.rows) continue;
        if constexpr (LB == 1) {
            // One-digit operands.  A wave issues at most one instruction every 4 clocks, scalar ones
            // included, so the epilogue is priced in INSTRUCTIONS per element (the int64 form: 44, of
            // them 9 64-bit multiply-adds).  Here: fp64 arithmetic (qf_epi_f64, 15 instructions), and
            // output / addend addressed through buffer descriptors -- base = the block's first message
            // row at the row block's first column, records = the bytes of the messages that exist, so
            // the hardware range check IS the batch bound (loads beyond it return 0, stores are
            // dropped) and an element's address is one v_add of a scalar row offset to the lane's
            // offset (message 4h, column lane & 31).  No per-element branch, no 64-bit address math.
            constexpr bool BATCH = !PACKED;
            const uint64_t mu0 = (mb0 + i) * 32, col0 = (uint64_t)(rb0 + j) * 32;
            const int64_t mleft = (int64_t)a.B - (int64_t)mu0;
            if (mleft <= 0) continue;
            const uint32_t mlim = mleft > 32 ? 32u : (uint32_t)mleft;            // messages of this block inside the batch
            const uint32_t nrec = mlim * rowbytes - (uint32_t)col0 * 4;
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + mu0 * a.rows + col0), 0, (int)nrec, 0x00020000);
            const auto rs_add = __builtin_amdgcn_make_buffer_rsrc((void *)(a.addend + mu0 * a.rows + col0), 0, a.addend ? (int)nrec : 0, 0x00020000);
            const bool has_last = a.msg != nullptr && lastrow - (rb0 + j) * 32 < 32;
            uint32_t add[BATCH ? 16 : 1];
            if constexpr (BATCH) {
                if (a.addend) {          // 16 loads in flight
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        add[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)(voff_st + ((r & 3) + 8 * (r >> 2)) * rowbytes), 0, QF_EPI_AUX_LD);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) add[r] = 0;
                }
            }
#ifdef QF_STAMP
            QF_T(te0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QF_T(te1)
            QF_ACC(s_eld, te0, te1)
#endif
            if constexpr (BATCH)
                if (has_last) {   // the one row block with the last key row: message term folded into the addend words (< 2^32)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        add[r] += (krow == lastrow && ((msgmask[i] >> ((r & 3) + 8 * (r >> 2) + 4 * he)) & 1)) ? a.mod.qhalf : 0u;
                }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int fr = (r & 3) + 8 * (r >> 2);
                const int voff = (int)(voff_st + fr * rowbytes);
#ifdef QF_ABL_NOEPI
                if (acc[i][j][0][r] == 0x7fffffff) __builtin_amdgcn_raw_buffer_store_b32(acc[i][j][S > 1 ? 1 : 0][r] + acc[i][j][S - 1][r], rs_out, voff, 0, 0);
                continue;
#endif
                double extra;
                if constexpr (BATCH) extra = (double)add[r];
                else extra = (double)(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_add, voff, 0, 0);
                if constexpr (!BATCH)
                    if (has_last) {   // uniform: only the row block that holds the last key row
                        asm volatile("");
                        if (krow == lastrow && ((msgmask[i] >> (fr + 4 * he)) & 1)) extra += (double)a.mod.qhalf;
                    }
                __builtin_amdgcn_raw_buffer_store_b32(
                    (int)qf_epi_f64<S>(acc[i][j][0][r], acc[i][j][S > 1 ? 1 : 0][r], acc[i][j][S > 2 ? 2 : 0][r], acc[i][j][S > 3 ? 3 : 0][r], extra, a),
                    rs_out, voff, 0, QF_EPI_AUX_ST);
            }
#ifdef QF_STAMP
            QF_T(te2)
            QF_ACC(s_ecomp, te1, te2)
#endif
        } else {
        uint32_t add[16];
        const bool late_addend = a.addend != nullptr;
        if (late_addend) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint64_t mrow = (mb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                mrow = mrow < a.B ? mrow : a.B - 1;
                add[r] = a.addend[mrow * a.rows + krow];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) add[r] = 0;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint64_t mrow = (mb0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow >= a.B) continue;
            int64_t x = 0;
#pragma unroll
            for (int s = 0; s < S; s++) x += (int64_t)acc[i][j][s][r] * (int64_t)a.cs[s];
            x += (int64_t)add[r];
            if (krow == lastrow && ((msgmask[i] >> ((r & 3) + 8 * (r >> 2) + 4 * h)) & 1)) x += (int64_t)a.mod.qhalf;
            a.out[mrow * a.rows + krow] = qf_red_i64(x, a.mod);
        }
        }
    }
    }
#ifdef QF_STAMP
    QF_T(te3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QF_T(tl2)
    QF_ACC(s_edrain, te3, tl2)
    QF_ACC(s_epi, tl1, tl2)
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * QF_NW + wave) * 8;
        d[0] = s_loop; d[1] = s_vm; d[2] = s_bar; d[3] = s_at; d[4] = s_epi; d[5] = tl0; d[6] = tl2;
        unsigned long long *d2 = a.dbg + (size_t)gridDim.x * gridDim.y * QF_NW * 8 + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * QF_NW + wave) * 4;
        d2[0] = s_eld; d2[1] = s_ecomp; d2[2] = s_edrain; d2[3] = tl1;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Bit operands on v_mfma_i32_16x16x64_i8 (round 2).  Same tiles, same HBM and LDS images, same
// LDS-DMA pipeline as qf_gemm_body's pinned loop -- only the matrix instruction changes: a 32x32
// output block over 64 columns is four 16x16x64 MFMAs (two message halves x two row halves)
// instead of two 32x32x32.  The chip holds ~2.3 GHz under 16x16x64 where it drops to 1.7-1.8 GHz
// under 32x32x32 on the same operands (tools/ubench_mfma.hip), and the full kernel is 14-16 %
// faster for it.  Nothing is re-laid-out: lane l of a 16x16x64 operand wants row (l & 15), columns
// 16*(l >> 4) .. +15 of the 64 -- the SAME 16-byte chunk some lane of a 32x32x32 fragment holds
// (rows 16*half + (l & 15), k-half (l >> 4) & 1, of k-step 2*s' + (l >> 5)) -- so every lane
// simply reads a different chunk of the 1-KiB fragment images, and the bits of a message row are
// expanded per 16-message half.
// PL (round 3): the same body for DIGIT-PLANE batch operands (one digit per k-step: LPR's folded r, SIS with Z_q
// inputs).  The planes stay in the 32x32x32 fragment order the recode kernel writes; as with the key fragments,
// every lane simply loads a different 16-byte chunk: message row 16*mg + (l & 15), k-half (l >> 4) & 1 of k-step
// 2*sp + (l >> 5) -- four 16-byte loads per lane and k-tile (two message halves x two k-steps of 64), nothing to
// expand.  The point is the clock: under v_mfma_i32_32x32x32_i8 the LPR launch ran at 1.67 GHz (power-managed).
template <int LA, int WN, int PK, bool PL = false>
__device__ __forceinline__ void qf_gemm_body16(const QfGemmArgs &a, const uint32_t bx, const uint32_t rb_first) {
    static_assert(!PL || PK == 0, "digit-plane operands: no packed remainder block");
    constexpr bool PACKED = PK != 0;
    constexpr bool PK16 = PK == 2;      // the packed block's r*LA digit rows all lie in its first 16: no second row half
    constexpr int S = LA;                                       // one operand digit
    constexpr int NBF = PACKED ? (WN - 1) * LA + 1 : WN * LA;   // key fragments per 32-wide k-step
    constexpr int FRAGS = NBF * QF_KS;
    constexpr int TILE_BYTES = FRAGS * 1024;
    constexpr int NSR = (FRAGS + QF_NW - 1) / QF_NW;
    constexpr int H = 2 * NBF - (PK16 ? 1 : 0);                 // groups per k-step of 64: NBF fragments x 2 row halves
    constexpr int G = 2 * H;                                    // groups (= LDS reads) per k-tile
    constexpr int ACCREGS = (PACKED ? (WN - 1) * S * 16 + (PK16 ? 8 : 16) : WN * S * 16);
    constexpr int D0 = (NBF >= 2 && (ACCREGS < 192 || (PL && QF_PL_D4)) && !PK16) ? 4 : 2;   // chunks in flight (ring registers / 4)
    constexpr int D = D0 <= G / 2 ? D0 : G / 2;
    static_assert(G % D == 0 && D <= G / 2, "ring slots repeat per tile; reads must not pass the mid-tile barrier");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 3 * TILE_BYTES

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const uint32_t rb0 = rb_first;
    const uint64_t mb0 = (uint64_t)bx * QF_NW + wave;           // this wave's message block (32 messages)
    const uint32_t ntiles = a.ks / QF_KS;
    uint32_t msgmask = 0;             // message bits of the wave's 32 messages (bit = message row inside the block)

    v4i acc[2][WN][2][S];             // [message half][row block][row half][digit]
    {   // ---- main part (its lane-derived values end with it)
    const int lane = (int)threadIdx.x & 63;
    const int lc = lane & 15;         // row inside a group of 16 (message row as A, key row as B / D column)
    const int lq = lane >> 4;         // k-quarter this lane feeds; message quarter (4 rows) it receives
#pragma unroll
    for (int mg = 0; mg < 2; mg++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int kg = 0; kg < 2; kg++)
#pragma unroll
                for (int s = 0; s < S; s++) acc[mg][j][kg][s] = v4i{0, 0, 0, 0};

    // batch operand: the 128 bits per k-tile of message rows 16*mg + lc (inline-asm loads, see QF_LOAD_AT),
    // addressed as (uniform base of the workgroup's first message row) + (32-bit per-lane offset)
    // Row-major bits (QF_X_BITS): a row's k-tiles are consecutive 16-byte words, rows a.xstride apart -- a wave's request for its
    // 32 rows touches 32 different 128-byte lines.  K-tile-major bits (QF_X_BITS_KT): the words of one k-tile are consecutive over
    // the rows, planes a.xstride apart -- the same request is 512 contiguous bytes (4 lines): the CU's vector L1 moves 52 instead
    // of 80 lines per wave and k-tile, Regev cfg2 / SIS 3.5-4 % faster (profiles/r3/gemm_ablation.log).
    const uint64_t tile_row0 = (uint64_t)bx * QF_NW * 32;
    const size_t xrowb = a.xkt ? (size_t)16 : a.xstride;            // bytes between consecutive rows
    const size_t xkt_s = a.xkt ? a.xstride : (size_t)0;             // per k-tile: advance of the uniform base ...
    const uint32_t xkt_v = a.xkt ? 0u : 16u;                        // ... and of the per-lane offset,
    const size_t xkt_vlim = a.xkt ? (size_t)0 : a.xstride - 16;     // which stays inside the row
    const uint8_t *xbase = PL ? nullptr : a.xbits + (tile_row0 < a.B ? tile_row0 : a.B - 1) * xrowb;
    uint32_t xoff[2] = {0, 0};
    if constexpr (!PL) {
#pragma unroll
    for (int mg = 0; mg < 2; mg++) {
        const uint64_t mrow_l = mb0 * 32 + 16 * mg + lc;
        const uint64_t mrow_c = mrow_l < a.B ? mrow_l : a.B - 1;
        xoff[mg] = (uint32_t)((mrow_c - (tile_row0 < a.B ? tile_row0 : a.B - 1)) * xrowb);   // < 256 rows
    }
    }
    v4i at[2][2];
    // digit planes: the wave's message block is ks * xlb_total consecutive KiB; per-lane chunk as described above
    v4i atp[PL ? 2 : 1][2][2];                         // [set][message half][k-step of 64]
    const int8_t *pbase = nullptr;
    uint32_t poff[2] = {0, 0};
    const uint32_t pstep = a.xlb_total * 1024;         // bytes between consecutive k-steps of a block
    if constexpr (PL) {
        const uint64_t nmb = (a.B + 31) / 32;
        const uint64_t mb_c = mb0 < nmb ? mb0 : nmb - 1;
        pbase = a.xplanes + ((mb_c * a.ks * a.xlb_total) + a.xlb_off) * 1024;
#pragma unroll
        for (int mg = 0; mg < 2; mg++) poff[mg] = (uint32_t)(lq >> 1) * pstep + (uint32_t)((16 * mg + lc) + 32 * (lq & 1)) * 16;
    }
#define QF16_LOAD_AT(set_, kt_)                                                                   \
    if constexpr (PL) {                                                                           \
        _Pragma("unroll") for (int mg_ = 0; mg_ < 2; mg_++)                                       \
            _Pragma("unroll") for (int sp_ = 0; sp_ < 2; sp_++)                                   \
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(atp[set_][mg_][sp_])         \
                             : "v"(poff[mg_] + ((uint32_t)(kt_) * QF_KS + 2 * sp_) * pstep), "s"(pbase) : "memory"); \
    } else {                                                                                      \
    const uint8_t *xb_ = xbase + (size_t)(kt_) * xkt_s;                                           \
    size_t off_ = (size_t)(kt_) * xkt_v;                                                          \
    off_ = off_ <= xkt_vlim ? off_ : xkt_vlim;                                                    \
    _Pragma("unroll") for (int mg_ = 0; mg_ < 2; mg_++) {                                         \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(at[set_][mg_]) : "v"(xoff[mg_] + (uint32_t)off_), "s"(xb_) : "memory"); \
    } }
#define QF16_WAIT_AT(set_, n_)                                                                    \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory");                                     \
    if constexpr (PL) {                                                                           \
        _Pragma("unroll") for (int mg_ = 0; mg_ < 2; mg_++)                                       \
            _Pragma("unroll") for (int sp_ = 0; sp_ < 2; sp_++) asm volatile("" : "+v"(atp[set_][mg_][sp_])); \
    } else {                                                                                      \
    asm volatile("" : "+v"(at[set_][0]));                                                         \
    asm volatile("" : "+v"(at[set_][1])); }
    // 16 bits -> 16 bytes for k-step sp (of 64) of the tile whose words are in at[set_][mg_]: the lane's
    // bits are half (lq & 1) of dword 2*sp + (lq >> 1)
    // the lane's 16 bits of the k-step: word (lq >> 1) of the step's two, half (lq & 1) of it.  The two
    // candidate words pass through an empty asm: hipcc otherwise turns "cond ? v[3] : v[2]" into an
    // element extract with a PER-LANE index and lowers that as a compare/select chain over all 16 dwords
    // of at[][] (68 VALU instructions + hazard nops per k-tile instead of 4 selects).
#define QF16_WSEL(set_, mg_, sp_) {                                                               \
        uint32_t w0_ = (uint32_t)at[set_][mg_][2 * (sp_)], w1_ = (uint32_t)at[set_][mg_][2 * (sp_) + 1]; \
        asm volatile("" : "+v"(w0_), "+v"(w1_));                                                  \
        const uint32_t w_ = hiword ? w1_ : w0_;                                                   \
        wsel[mg_] = __builtin_amdgcn_ubfe(w_, halfsh, 16); }
#define QF16_EXPAND(dst_, mg_, d_) dst_[d_] = (int)((((wsel[mg_] >> (4 * (d_))) & 0xFu) * 0x00204081u) & 0x01010101u);

    QF16_LOAD_AT(0, 0)
    QF_STAGE_SETUP()
    QF_STAGE(0, 0)
    QF_STAGE(ntiles > 1 ? 1 : 0, 1)
    if (a.msg && (uint64_t)(rb0 + WN) * 32 >= a.rows) {
        const uint64_t mrow_l = mb0 * 32 + (lane & 31);
        const bool bit = lane < 32 && mrow_l < a.B && (a.msg[mrow_l < a.B ? mrow_l : 0] & 1);
        msgmask = (uint32_t)__ballot(bit);
    }
    QF16_WAIT_AT(0, 0)
    __syncthreads();  // drains the DMA (vmcnt(0)) and synchronises the workgroup

    // per-lane chunk of a fragment image: row lc (+16 per row half, in the read's immediate offset),
    // k-half (lq & 1), of k-step 2*sp + (lq >> 1) -- consecutive k-steps of a fragment are LA KiB apart
    // (1 KiB in the packed block)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t laneA = lds0 + (uint32_t)(lc + 32 * (lq & 1)) * 16 + (uint32_t)(lq >> 1) * LA * 1024;
    const uint32_t laneP = lds0 + (uint32_t)(lc + 32 * (lq & 1)) * 16 + (uint32_t)(lq >> 1) * 1024;
#define QF16_OFF(fi_, sp_, kg_) (((PACKED && (fi_) == NBF - 1) ? ((WN - 1) * QF_KS * LA + 2 * (sp_)) : ((((fi_) / LA) * QF_KS + 2 * (sp_)) * LA + (fi_) % LA)) * 1024 + (kg_) * 256)
    // group g of a tile -> k-step of 64 (g / H), fragment ((g % H) / 2) and row half ((g % H) % 2)
#define QF16_RD(slot_, tA_, tP_, g_)                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[slot_])                              \
                 : "v"((PACKED && ((g_) % H) / 2 == NBF - 1) ? (tP_) : (tA_)),                    \
                   "n"(QF16_OFF(((g_) % H) / 2, (g_) / H, ((g_) % H) % 2)));
    v4i ring[D];
    v4i afb[2][2];                    // [k-step of 64 (parity)][message half]
    uint32_t wsel[2];
    const bool hiword = (lq >> 1) != 0;
    const uint32_t halfsh = (uint32_t)(lq & 1) * 16;
#pragma unroll
    for (int g = 0; g < D; g++) QF16_RD(g, laneA, laneP, g)
    if constexpr (!PL) {
#pragma unroll
    for (int mg = 0; mg < 2; mg++) {
        QF16_WSEL(0, mg, 0)
#pragma unroll
        for (int d = 0; d < 4; d++) QF16_EXPAND(afb[0][mg], mg, d)
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    // group of the tile's second k-step behind which the next tile's bits are waited for and expanded
    // (digit planes: nothing to expand, the wait sits in front of the tile's last group)
    constexpr int WPOS = PL ? G - 1 : (H + NBF < G ? H + NBF : G - 1);
    constexpr bool SPREAD = (WPOS - H) + 10 <= H;   // one expansion piece per group (10 pieces per k-step of 64), else all at once
    uint32_t buf = 0;
    for (uint32_t kt = 0; kt < ntiles; kt++) {
        const uint32_t buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
        const uint32_t tA = laneA + buf * TILE_BYTES, tP = laneP + buf * TILE_BYTES;
        const uint32_t kt1 = kt + 1 < ntiles ? kt + 1 : kt;  // clamped: the tail prefetches are discarded
        const uint32_t kt2 = kt + 2 < ntiles ? kt + 2 : ntiles - 1;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int sp = g / H, rem = g % H, fi = rem / 2, kg = rem % 2, slot = g % D;
#ifndef QF_ABL_NOPRIO
            // the two waves of a SIMD take turns at the higher issue priority (see qf_gemm_body)
            if (g == 0) { if (wave >= QF_NW / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
            if (g == H) { if (wave >= QF_NW / 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            if (g == H) {
                // mid-tile: tile t+1 has landed for everyone, tile t-1's buffer is free
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                QF16_LOAD_AT(1, kt1)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == WPOS) {
                // the next tile's bits: all but the DMA pieces issued behind the load have retired
                constexpr int P2 = qf16_pieces_before(NSR, H, WPOS);
                QF16_WAIT_AT(1, P2)
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(D - 1));
            asm volatile("" : "+v"(ring[slot]));
            const int j = (PACKED && fi == NBF - 1) ? WN - 1 : fi / LA;
            const int l = (PACKED && fi == NBF - 1) ? 0 : fi % LA;
#pragma unroll
            for (int mg = 0; mg < 2; mg++)
                acc[mg][j][kg][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(PL ? atp[0][mg][sp] : afb[sp][mg], ring[slot], acc[mg][j][kg][l], 0, 0, 0);
            const int gn = g + D;
            if (gn < G) { QF16_RD(slot, tA, tP, gn) }
            else { QF16_RD(slot, laneA + buf1 * TILE_BYTES, laneP + buf1 * TILE_BYTES, gn - G) }   // the next tile's first chunks
            // the operand of the NEXT k-step of 64 (sp = 0: this tile's second; sp = 1, behind the counted
            // wait: the next tile's first): per message half one word select + four dword expansions
            if constexpr (!PL) {
                const int e0 = sp == 0 ? 0 : WPOS - H;
#pragma unroll
                for (int mg = 0; mg < 2; mg++) {
                    if (SPREAD ? rem == e0 + 5 * mg : rem == e0) {
                        if (sp == 0) { QF16_WSEL(0, mg, 1) } else { QF16_WSEL(1, mg, 0) }
                    }
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        if (SPREAD ? rem == e0 + 5 * mg + 1 + d : rem == e0) { QF16_EXPAND(afb[sp ^ 1][mg], mg, d) }
                }
            }
            if (g >= H) {
#pragma unroll
                for (int p = 0; p < NSR; p++)
                    if (qf16_piece_group(p, NSR, H) == g) { QF_STAGE_PART(kt2, buf2, p, p + 1) }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PL) {
#pragma unroll
            for (int mg = 0; mg < 2; mg++)
#pragma unroll
                for (int sp = 0; sp < 2; sp++) atp[0][mg][sp] = atp[1][mg][sp];
        } else {
        at[0][0] = at[1][0];
        at[0][1] = at[1][1];
        }
        buf = buf1;
    }
    // the reads issued for a tile past the end must not land in registers the epilogue re-uses
    asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
    for (int g = 0; g < D; g++) asm volatile("" : "+v"(ring[g]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must not outlive the workgroup's LDS
    __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue.  D tile of a 16x16x64 MFMA: lane holds key row 16*kg + lc of the block, messages
    // 16*mg + 4*lq + r (r = register 0..3).  The lane index is re-derived here (mbcnt, opaque to the
    // compiler): otherwise hipcc computes the epilogue's per-lane addresses in front of the main loop
    // and keeps them alive -- or spills them -- across it.
    uint32_t lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int lane = (int)lane_e, lc = lane & 15, lq = lane >> 4;
    const uint32_t lastrow = a.rows - 1;
    const uint32_t rowbytes = a.rows * 4;
    const uint32_t voff = (uint32_t)(4 * lq) * rowbytes + (uint32_t)lc * 4;
    const int64_t mleft = (int64_t)a.B - (int64_t)(mb0 * 32);
    if (mleft <= 0) return;                                       // uniform; nothing after the epilogue
    const uint32_t mlim = mleft > 32 ? 32u : (uint32_t)mleft;     // messages of this wave's block inside the batch
    // Addend words of a whole message half -- WN blocks x 8 per lane -- requested together, ahead of the half's
    // arithmetic (digit-plane operands; 8 at a time left the LPR epilogue 0.1 ms behind the 32x32x32 body's 16 at a
    // time).  The second half's words are requested as soon as the first half's first row block is done and its
    // accumulator registers are free (>= 8 LA per block against 8 WN words), so their latency runs under the
    // rest of the first half.  A descriptor with 0 records returns zeros without touching memory.
    constexpr bool PRE = PL && WN * LA * 16 + WN * 8 <= 216;   // tiles of 192 accumulator registers + 32 words: no room, 8 at a time
    uint32_t addall[PRE ? 2 : 1][PRE ? WN : 1][2][4];
#define QF16_ADD_HALF(MG_)                                                                                                \
    if constexpr (PRE) {                                                                                                  \
        if (a.addend) {                                                                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++) {                                                           \
                const bool live_ = rb0 + j_ < a.rb;                                                                       \
                const uint64_t c0_ = live_ ? (uint64_t)(rb0 + j_) * 32 : 0;                                               \
                const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)(a.addend + mb0 * 32 * a.rows + c0_), 0,      \
                                                                    live_ ? (int)(mlim * rowbytes - (uint32_t)c0_ * 4) : 0, 0x00020000); \
                _Pragma("unroll") for (int kg_ = 0; kg_ < 2; kg_++)                                                       \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++)                                                      \
                        addall[PRE ? (MG_) : 0][PRE ? j_ : 0][kg_][r_] =                                                  \
                            (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_a, (int)(voff + (16 * (MG_) + r_) * rowbytes) + 64 * kg_, 0, 0); \
            }                                                                                                             \
        } else {                                                                                                          \
            _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                                             \
                _Pragma("unroll") for (int kg_ = 0; kg_ < 2; kg_++)                                                       \
                    _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) addall[PRE ? (MG_) : 0][PRE ? j_ : 0][kg_][r_] = 0;  \
        }                                                                                                                 \
    }
    QF16_ADD_HALF(0)
    if constexpr (WN == 1) { QF16_ADD_HALF(1) }
#pragma unroll
    for (int mg = 0; mg < 2; mg++) {
#pragma unroll
    for (int j = 0; j < WN; j++) {
        if constexpr (WN > 1) { if (mg == 0 && j == 1) { QF16_ADD_HALF(1) } }
        if (PACKED && j == WN - 1) {
            // remainder block: fragment row p = i*LA + l holds digit l of trailing row i, i.e. it sits in row
            // half p >> 4 at lane column p & 15 (same message quarter).  Lane column c serves trailing rows
            // c and c + 16: it gathers their LA digits from the lanes that hold them.
#pragma unroll
            for (int hh = 0; hh < (PK16 ? 1 : 2); hh++) {
                const uint32_t ir = (uint32_t)lc + 16 * hh;
                const uint32_t krow = (rb0 + j) * 32 + ir;
                const bool owner = ir < a.rem_rows;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint64_t mrow = mb0 * 32 + 16 * mg + 4 * lq + r;
                    int64_t x = 0;
#pragma unroll
                    for (int l = 0; l < LA; l++) {
                        const uint32_t pp = (ir * LA + l) & 31;          // rows beyond the block: masked, never owners
                        const int src = (int)((pp & 15) | (uint32_t)(lane & 48));
                        int v = __shfl(acc[mg][j][0][0][r], src, 64);
                        if constexpr (!PK16) { const int v1 = __shfl(acc[mg][j][1][0][r], src, 64); v = (pp >> 4) ? v1 : v; }
                        x += (int64_t)v * (int64_t)a.cs[l];
                    }
                    if (!owner || mrow >= a.B) continue;
                    if (a.addend) x += (int64_t)a.addend[mrow * a.rows + krow];
                    if (krow == lastrow && ((msgmask >> (16 * mg + 4 * lq + r)) & 1)) x += (int64_t)a.mod.qhalf;
                    a.out[mrow * a.rows + krow] = qf_red_i64(x, a.mod);
                }
            }
            continue;
        }
        // plain blocks: fp64 arithmetic, buffer descriptors whose record count is the batch bound (see
        // qf_gemm_body's epilogue); lane offset = (message 4*lq, column lc), element = + (16*mg + r) rows,
        // + 64 bytes for the second row half
        if (rb0 + j >= a.rb) continue;
        const uint64_t col0 = (uint64_t)(rb0 + j) * 32;
        const uint32_t nrec = mlim * rowbytes - (uint32_t)col0 * 4;
        const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + mb0 * 32 * a.rows + col0), 0, (int)nrec, 0x00020000);
        const auto rs_add = __builtin_amdgcn_make_buffer_rsrc((void *)(a.addend + mb0 * 32 * a.rows + col0), 0, a.addend ? (int)nrec : 0, 0x00020000);
        const bool has_last = a.msg != nullptr && lastrow - (rb0 + j) * 32 < 32;
        if (a.addend == nullptr && !has_last && !a.epi2) {
            // nothing to add and the one-step recombination is exact (Regev, SIS: all but one row block):
            // the three wave-uniform tests leave the element code -- 13 instructions per element instead of 22
            asm volatile("");
#pragma unroll
            for (int kg = 0; kg < 2; kg++) {
                if ((rb0 + j) * 32 + 16 * kg + lc >= a.rows) continue;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    __builtin_amdgcn_raw_buffer_store_b32(
                        (int)qf_epi_f64_plain<S>(acc[mg][j][kg][0][r], acc[mg][j][kg][S > 1 ? 1 : 0][r], acc[mg][j][kg][S > 2 ? 2 : 0][r],
                                                 acc[mg][j][kg][S > 3 ? 3 : 0][r], a),
                        rs_out, (int)(voff + (16 * mg + r) * rowbytes) + 64 * kg, 0, 0);
            }
            continue;
        }
        // the block's 8 addend words per lane are requested together (one at a time, each load was waited for on its
        // own: LPR 2.18 -> 2.51 ms when the digit-plane path first came through this epilogue); a descriptor with 0
        // records (no addend) returns zeros without touching memory
        uint32_t addw[2][4];
#pragma unroll
        for (int kg = 0; kg < 2; kg++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (!PRE) addw[kg][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)(voff + (16 * mg + r) * rowbytes) + 64 * kg, 0, 0);
                else addw[kg][r] = addall[PRE ? mg : 0][PRE ? j : 0][kg][r];
            }
#pragma unroll
        for (int kg = 0; kg < 2; kg++) {
            const uint32_t krow = (rb0 + j) * 32 + 16 * kg + lc;
            if (krow >= a.rows) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int vo = (int)(voff + (16 * mg + r) * rowbytes) + 64 * kg;
                double extra = (double)addw[kg][r];
                if (has_last) {   // uniform: only the row block that holds the last key row
                    asm volatile("");
                    if (krow == lastrow && ((msgmask >> (16 * mg + 4 * lq + r)) & 1)) extra += (double)a.mod.qhalf;
                }
                __builtin_amdgcn_raw_buffer_store_b32(
                    (int)qf_epi_f64<S>(acc[mg][j][kg][0][r], acc[mg][j][kg][S > 1 ? 1 : 0][r], acc[mg][j][kg][S > 2 ? 2 : 0][r],
                                       acc[mg][j][kg][S > 3 ? 3 : 0][r], extra, a),
                    rs_out, vo, 0, 0);
            }
        }
    }
    }
}

// Raster order.  Plain: workgroups run message-tile-fastest, so one column tile's key planes stay
// in L2 while the whole batch streams by -- and the batch operand is re-read from HBM once per
// column tile.  With a.mchunk set, CHUNKS of a.mchunk message tiles run through every column tile
// before the next chunk starts: a chunk's operand tiles are re-used while still in L2 (workgroup
// ids go round-robin over the 8 XCDs, so with a chunk that is a multiple of 8 a message tile
// meets all its column tiles on the same XCD).  LPR n=1024 (537 MB of digit planes x 11 column
// tiles): 3.31 -> 2.82 ms.
__device__ __forceinline__ void qf_gemm_raster(const QfGemmArgs &a, uint32_t &bx, uint32_t &by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (a.mchunk) {
        const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;
        const uint32_t per = a.mchunk * gridDim.y;
        const uint32_t base = (lin / per) * a.mchunk, w = lin % per;
        const uint32_t sz = min(a.mchunk, gridDim.x - base);
        by = w / sz; bx = base + w % sz;
    }
}

template <int LA, int LB, int WN, int MW, bool BITS>
__global__ __launch_bounds__(QF_NW * 64, (MW * WN * (LA + LB - 1) * 16 + (BITS ? 16 : 8 * QF_KS * LB) <= 72 ? 4 : 2)) void qf_gemm_kernel(QfGemmArgs a) {
    uint32_t bx, by;
    qf_gemm_raster(a, bx, by);
#ifndef QF_GEMM_MFMA32
    if constexpr (BITS && MW == 1 && LB == 1) qf_gemm_body16<LA, WN, 0>(a, bx, by * WN);
    else
#ifndef QF_GEMM_PLANES32
    if constexpr (!BITS && MW == 1 && LB == 1) qf_gemm_body16<LA, WN, 0, true>(a, bx, by * WN);
    else
#endif
#endif
    qf_gemm_body<LA, LB, WN, MW, BITS, false>(a, bx, by * WN);
}
// bit operands on a key with a packed remainder block: the last column tile ends with it
template <int LA, int WN>
__global__ __launch_bounds__(QF_NW * 64, (WN * LA * 16 <= 48 ? 4 : 2)) void qf_gemm_packed_kernel(QfGemmArgs a) {
    uint32_t bx, by;
    qf_gemm_raster(a, bx, by);
#ifndef QF_GEMM_MFMA32
    if (by + 1 == gridDim.y) {
        if (a.rem_rows * LA <= 16) qf_gemm_body16<LA, WN, 2>(a, bx, by * WN);
        else qf_gemm_body16<LA, WN, 1>(a, bx, by * WN);
    } else qf_gemm_body16<LA, WN, 0>(a, bx, by * WN);
#else
    if (by + 1 == gridDim.y) qf_gemm_body<LA, 1, WN, 1, true, true>(a, bx, by * WN);
    else qf_gemm_body<LA, 1, WN, 1, true, false>(a, bx, by * WN);
#endif
}
// the same with the remainder block as an EXTRA fragment of the last column tile: the key's full row
// blocks split evenly into tiles of WN, the last tile computes WN blocks + the packed one (a body of
// WN+1 "blocks", the last of them one fragment wide).  Regev n = 256: two tiles of 12 and 13 fragments
// per k-step instead of three of 9, 9 and 7.
template <int LA, int WN>
__global__ __launch_bounds__(QF_NW * 64, 2) void qf_gemm_packedx_kernel(QfGemmArgs a) {
    uint32_t bx, by;
    qf_gemm_raster(a, bx, by);
#ifndef QF_GEMM_MFMA32
    if (by + 1 == gridDim.y) qf_gemm_body16<LA, WN + 1, 2>(a, bx, by * WN);   // dispatched only when rem_rows*LA <= 16
    else qf_gemm_body16<LA, WN, 0>(a, bx, by * WN);
#else
    if (by + 1 == gridDim.y) qf_gemm_body<LA, 1, WN + 1, 1, true, true>(a, bx, by * WN);
    else qf_gemm_body<LA, 1, WN, 1, true, false>(a, bx, by * WN);
#endif
}

// ---------------------------------------------------------------------------------------------
// Short inner dimension, many key rows (round 3): DualRegev -- c^t = s^t [A|u] + e^t with the key
// (m+1) x n = 5,314 x 230 at secure128 (dual_regev.rs:301-303,403), K' = 3 folded digits x 256 = 768.
// In the tiled kernel above a workgroup of that shape runs 6 k-tiles and then an epilogue of the same
// length, all 8 waves in step (they meet at a barrier every k-tile): the matrix pipe idles through
// every epilogue, the memory pipe through every main loop (stamps: loop 0.55, epilogue 0.45 of a
// wave's cycles; MFMA-busy 0.25; 1.33 ms for 65,536 encryptions).
// Here the roles of the loops are swapped.  A workgroup takes ONE column tile (WN*32 key rows) and
// keeps ALL its digit planes in LDS -- WN * K'/32 * LA KiB, loaded once (144 KiB for the shape
// above) -- and its 8 waves then sweep over message blocks on their own: no barrier after the
// prologue, so one wave's epilogue (addend loads, fp64 reduction, stores) runs beside the MFMAs of
// the other wave of its SIMD.  The batch operand comes from global memory a group of 4 k-steps
// ahead; the next block's first group and this block's addend words are requested before the
// stores of the epilogue are issued (vmcnt retires in order: a load queued behind 32 stores could
// not be consumed before they have all been acknowledged).
// Work units (column tile c, message split p) are dealt to the XCDs in runs: unit u = c*P + p goes
// to XCD u / upx, so the workgroups that share a stream of operand planes (same p, neighbouring c)
// and the output lines that straddle two column tiles meet in ONE L2.
// MW = message blocks (of 32) a wave multiplies per key fragment it reads from LDS.  One 1-KiB fragment read per
// 32-cycle MFMA and SIMD is exactly the LDS port's 128 B/clk/CU: at MW = 1 the matrix pipe and the LDS port run at
// the same limit (measured: the main loop alone reaches 0.63 of both).  <WN = 1, MW = 2> halves the LDS reads per
// MFMA at the same accumulator count (the operand planes are then read from L2 once per row block instead of
// once per two).
template <int LA, int WN, int MW>
__global__ __launch_bounds__(QF_NW * 64, 2) void qf_gemm_shortk_kernel(QfGemmArgs a, uint32_t P, uint32_t nunits, uint32_t upx) {
    constexpr int S = LA;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // WN * ks * LA KiB
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t u = xcd * upx + slot;
    if (slot >= upx || u >= nunits) return;                       // uniform: whole workgroup
    const uint32_t c = u / P, p = u % P;
    const uint32_t rb0 = c * WN;
    const uint64_t nmb = (a.B + 31) / 32;                         // message blocks
    const uint64_t nmp = (nmb + MW - 1) / MW;                     // groups of MW blocks: a wave's unit of work
    const uint64_t per = (nmp + P - 1) / P;
    const uint64_t mp_begin = p * per, mp_end = mp_begin + per < nmp ? mp_begin + per : nmp;
    const uint32_t fprb = a.ks * LA;                              // fragments (KiB) per row block

    // ---- the column tile's planes -> LDS, image [j][k-step][digit][64 lanes][16 B] = the HBM order of a row block
    for (uint32_t f = (uint32_t)wave; f < WN * fprb; f += QF_NW) {
        const uint32_t j = f / fprb, rem = f % fprb;
        uint32_t rbi = rb0 + j;
        rbi = rbi < a.rb ? rbi : a.rb - 1;                        // padded column tile: harmless duplicate
        const int8_t *g = a.kplanes + ((size_t)rbi * fprb + rem) * 1024 + (size_t)lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(smem + (size_t)f * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mp_begin + wave >= mp_end) return;

    const uint32_t lastrow = a.rows - 1;
    const uint32_t rowbytes = a.rows * 4;
    const uint32_t he = (uint32_t)lane >> 5;
    const uint32_t voff_st = 4 * he * rowbytes + ((uint32_t)lane & 31) * 4;   // byte offset of (message 4h, column lane & 31)
    const uint32_t G = a.ks / QF_KS;                              // groups of 4 k-steps
    const __attribute__((address_space(3))) uint8_t *ldsl = (const __attribute__((address_space(3))) uint8_t *)smem + lane * 16;
    const bool tile_has_last = a.msg != nullptr && lastrow >= rb0 * 32 && lastrow < (rb0 + WN) * 32;

    // Operand registers: av = the group of 4 k-steps being multiplied, avn = the next one (requested at the top
    // of the current group: the last group of a block requests the first group of the wave's next block);
    // bf = the key fragments of the current and the next k-step (LDS reads one k-step ahead, one read between
    // consecutive MFMA groups; the k-step behind a block's last one is k-step 0 again: the tile is the same for
    // every block, so the read-ahead simply wraps around).
#ifdef QF_ABL_SK_HOTX
#define QFS_MB(m_) ((m_) & 7)      /* ablation: every wave re-reads the same 8 blocks of operand planes (L1/L2-hot) */
#else
#define QFS_MB(m_) (m_)
#endif
    constexpr int NBF = WN * LA;
    constexpr bool PFB = MW * NBF * 16 + 2 * NBF * 4 + MW * 8 * QF_KS + MW * WN * 16 <= 216;   // LA = 4: no registers for the second fragment set
    v4i av[MW][QF_KS], avn[MW][QF_KS], bf[PFB ? 2 : 1][NBF];
    // block m of group mp_ (clamped to the last block of the batch: its results are dropped by the epilogue's record counts)
#define QFS_BLOCK(mp_, m_) (((mp_) * MW + (m_)) < nmb ? ((mp_) * MW + (m_)) : nmb - 1)
#define QFS_LOAD_GROUP(dst_, mp_, kg_)                                                             \
    _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++) {                                            \
        const int8_t *x_ = a.xplanes + ((size_t)QFS_MB(QFS_BLOCK(mp_, m_)) * a.ks + (size_t)(kg_) * QF_KS) * 1024 + (size_t)lane * 16; \
        _Pragma("unroll") for (int i_ = 0; i_ < QF_KS; i_++) dst_[m_][i_] = *(const v4i *)(x_ + i_ * 1024); \
    }
#define QFS_LOAD_B(set_, k_)                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                              \
        _Pragma("unroll") for (int l_ = 0; l_ < LA; l_++)                                          \
            bf[set_][j_ * LA + l_] = *(const __attribute__((address_space(3))) v4i *)(ldsl + ((size_t)j_ * fprb + (size_t)(k_) * LA + l_) * 1024);
#define QFS_GROUP(kg_)                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < QF_KS; i_++) {                                         \
        if constexpr (PFB) {                                                                       \
            uint32_t kn_ = (kg_) * QF_KS + i_ + 1;                                                 \
            kn_ = kn_ < a.ks ? kn_ : 0;                                                            \
            QFS_LOAD_B((i_ + 1) & 1, kn_)                                                          \
        } else {                                                                                   \
            QFS_LOAD_B(0, (kg_) * QF_KS + i_)                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < WN; j_++)                                          \
            _Pragma("unroll") for (int l_ = 0; l_ < LA; l_++)                                      \
                _Pragma("unroll") for (int m_ = 0; m_ < MW; m_++)                                  \
                    acc[m_][j_][l_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[m_][i_], bf[PFB ? (i_ & 1) : 0][j_ * LA + l_], acc[m_][j_][l_], 0, 0, 0); \
        if constexpr (PFB) {                                                                       \
            _Pragma("unroll") for (int t_ = 0; t_ < NBF; t_++) {                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x008, MW, 0);                                \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
    QFS_LOAD_GROUP(av, mp_begin + wave, 0)
    if constexpr (PFB) { QFS_LOAD_B(0, 0) }
    for (uint64_t mp = mp_begin + wave; mp < mp_end; mp += QF_NW) {
        v16i acc[MW][WN][LA];
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int l = 0; l < LA; l++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[m][j][l][r] = 0;
        for (uint32_t kg = 0; kg + 1 < G; kg++) {
            QFS_LOAD_GROUP(avn, mp, kg + 1)
            __builtin_amdgcn_sched_barrier(0);
            QFS_GROUP(kg)
#pragma unroll
            for (int m = 0; m < MW; m++)
#pragma unroll
                for (int i = 0; i < QF_KS; i++) av[m][i] = avn[m][i];
        }
        // ---- last group: the next block's first group and the addend words of the whole tile are requested
        // first and land under its MFMAs
        const uint64_t mpn = mp + QF_NW < mp_end ? mp + QF_NW : mp;   // no next block: a harmless re-read
        QFS_LOAD_GROUP(avn, mpn, 0)
        uint32_t add[MW][WN][16], mlim[MW], msgmask[MW];
        uint64_t mu0[MW];
#pragma unroll
        for (int m = 0; m < MW; m++) {
            mu0[m] = (mp * MW + m) * 32;
            const int64_t mleft = (int64_t)a.B - (int64_t)mu0[m];
            mlim[m] = mleft > 32 ? 32u : (mleft > 0 ? (uint32_t)mleft : 0u);    // messages of this block inside the batch (0: a block past the end)
            if (mlim[m] == 0) mu0[m] = 0;                                          // keeps the descriptor bases inside the buffers
#pragma unroll
            for (int j = 0; j < WN; j++) {
                const uint64_t col0 = (uint64_t)(rb0 + j) * 32;
                const bool live = rb0 + j < a.rb && mlim[m] != 0;
                const uint32_t nrec = live ? mlim[m] * rowbytes - (uint32_t)col0 * 4 : 0;
                const auto rs_add = __builtin_amdgcn_make_buffer_rsrc((void *)(a.addend + mu0[m] * a.rows + (live ? col0 : 0)), 0, a.addend ? (int)nrec : 0, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    add[m][j][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_add, (int)(voff_st + ((r & 3) + 8 * (r >> 2)) * rowbytes), 0, 0);
            }
            msgmask[m] = 0;
            if (tile_has_last) {
                const uint64_t mrow_l = mu0[m] + ((uint32_t)lane & 31);
                const bool bit = lane < 32 && mlim[m] != 0 && mrow_l < a.B && (a.msg[mrow_l < a.B ? mrow_l : 0] & 1);
                msgmask[m] = (uint32_t)__ballot(bit);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        QFS_GROUP(G - 1)
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int i = 0; i < QF_KS; i++) av[m][i] = avn[m][i];
        // ---- epilogue (as the one-digit path of qf_gemm_body: fp64, buffer descriptors whose record count is the batch bound)
#pragma unroll
        for (int m = 0; m < MW; m++) {
#pragma unroll
        for (int j = 0; j < WN; j++) {
            const uint32_t krow = (rb0 + j) * 32 + ((uint32_t)lane & 31);
            if (rb0 + j >= a.rb || krow >= a.rows || mlim[m] == 0) continue;
            const uint64_t col0 = (uint64_t)(rb0 + j) * 32;
            const uint32_t nrec = mlim[m] * rowbytes - (uint32_t)col0 * 4;
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + mu0[m] * a.rows + col0), 0, (int)nrec, 0x00020000);
            const bool has_last = a.msg != nullptr && lastrow - (rb0 + j) * 32 < 32;
            if (has_last) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    add[m][j][r] += (krow == lastrow && ((msgmask[m] >> ((r & 3) + 8 * (r >> 2) + 4 * he)) & 1)) ? a.mod.qhalf : 0u;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int voff = (int)(voff_st + ((r & 3) + 8 * (r >> 2)) * rowbytes);
#ifdef QF_ABL_SK_NOEPI
                if (acc[m][j][0][r] == 0x7fffffff) __builtin_amdgcn_raw_buffer_store_b32(acc[m][j][S > 1 ? 1 : 0][r] + acc[m][j][S - 1][r] + (int)add[m][j][r], rs_out, voff, 0, 0);
                continue;
#endif
#ifdef QF_ABL_SK_NOSTORE
                { const int v_ = (int)qf_epi_f64<S>(acc[m][j][0][r], acc[m][j][S > 1 ? 1 : 0][r], acc[m][j][S > 2 ? 2 : 0][r], acc[m][j][S > 3 ? 3 : 0][r], (double)add[m][j][r], a);
                  if (v_ == 0x7fffffff) __builtin_amdgcn_raw_buffer_store_b32(v_, rs_out, voff, 0, 0); }
                continue;
#endif
                __builtin_amdgcn_raw_buffer_store_b32(
                    (int)qf_epi_f64<S>(acc[m][j][0][r], acc[m][j][S > 1 ? 1 : 0][r], acc[m][j][S > 2 ? 2 : 0][r], acc[m][j][S > 3 ? 3 : 0][r], (double)add[m][j][r], a),
                    rs_out, voff, 0, 0);
            }
        }
        }
    }
#undef QFS_MB
#undef QFS_BLOCK
#undef QFS_LOAD_GROUP
#undef QFS_LOAD_B
#undef QFS_GROUP
}

// units = column tiles x P message splits; P chosen so that the rounds of 256 workgroups (one per CU: the key
// tile takes most of a CU's LDS) are as full as possible while a unit keeps enough message blocks to amortise
// loading its tile
template <int LA, int WN, int MW>
static int qf_gemm_dispatch_shortk(qf_ctx *ctx, const QfGemmArgs &args) {
    const size_t lds = (size_t)WN * args.ks * LA * 1024;
    static QfPerDevice attr;
    QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_gemm_shortk_kernel<LA, WN, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
    const uint32_t nct = (args.rb + WN - 1) / WN;
    const uint64_t nmp = ((args.B + 31) / 32 + MW - 1) / MW;
    int ncu = 256;
    if (const char *e = getenv("QF_GEMM_SHORTK_CUS")) ncu = atoi(e) > 0 ? atoi(e) : ncu;
    uint32_t bestP = 1;
    double best = 1e300;
    for (uint32_t P = 1; P <= 512; P++) {
        const uint64_t per = (nmp + P - 1) / P;
        if (P > 1 && per < 4 * QF_NW) break;                      // at least 4 iterations per wave and unit
        const uint64_t units = (uint64_t)nct * P;
        const double cost = (double)((units + ncu - 1) / ncu) * ((double)((per + QF_NW - 1) / QF_NW) + 1.5);   // + ~1.5 iterations to load the tile
        if (cost < best) { best = cost; bestP = P; }
    }
    if (const char *e = getenv("QF_GEMM_SHORTK_P")) bestP = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : bestP;
    const uint32_t nunits = nct * bestP;
    const uint32_t upx = (nunits + 7) / 8;
    QfTimer t(ctx, &ctx->ms_gemm);
    hipLaunchKernelGGL((qf_gemm_shortk_kernel<LA, WN, MW>), dim3(upx * 8), dim3(QF_NW * 64), lds, ctx->stream, args, bestP, nunits, upx);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
// whether the short-K sweep serves a folded one-digit product: the tile's planes must fit a CU's LDS, and there
// must be enough rows and messages for the sweep to pay (QF_GEMM_SHORTK=0/1 overrides: tests, A/B).  Returns 0
// (no), 2 (<WN = 2, MW = 1>) or 1 (<WN = 1, MW = 2>: QF_GEMM_SHORTK_MW=2).
static int qf_gemm_shortk_wn(const QfGemmArgs &p, int la) {
    const uint64_t kib = (uint64_t)p.ks * la;
    int wn = 2 * kib <= 144 ? 2 : 0;
    const uint64_t nmb = (p.B + 31) / 32;
    bool use = wn && p.rb >= 16 && nmb >= 256;
    if (const char *e = getenv("QF_GEMM_SHORTK")) use = wn && atoi(e) != 0;
    if (use && la <= 3) if (const char *e = getenv("QF_GEMM_SHORTK_MW")) if (atoi(e) == 2) wn = 1;   // 4 key digits: 128 accumulator registers + two operand sets spill
    return use ? wn : 0;
}

// message tiles per raster chunk (see qf_gemm_raster); QF_GEMM_MCHUNK / QF_GEMM_MCHUNK_BITS override
static uint32_t qf_gemm_mchunk(bool bits, uint64_t nmsgtiles, unsigned coltiles) {
    // digit planes: always (LPR 3.31 -> 2.82 ms, SIS with Z_q inputs 4.35 -> 3.98 ms); bit operands
    // only on batches too large for the Infinity Cache (SIS B = 2^20: +2 %; Regev B = 65,536: -1 %)
    uint64_t chunk = bits ? (nmsgtiles >= 1024 ? 32 : 0) : 32;
    if (const char *e = getenv(bits ? "QF_GEMM_MCHUNK_BITS" : "QF_GEMM_MCHUNK")) chunk = (uint64_t)atoll(e);
    if (coltiles > 1 && chunk && chunk < nmsgtiles && nmsgtiles * coltiles < 0x7fffffffull) return (uint32_t)chunk;
    return 0;
}

template <int LA, int LB, int WN, int MW, bool BITS>
static int qf_gemm_dispatch_wn(qf_ctx *ctx, const QfGemmArgs &args) {
    constexpr int FRAGS = WN * QF_KS * LA;
    size_t lds = 3 * (size_t)FRAGS * 1024;
    static QfPerDevice attr;
    QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_gemm_kernel<LA, LB, WN, MW, BITS>,
                                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
    uint64_t nmsgtiles = (args.B + 32 * QF_NW * MW - 1) / (32 * QF_NW * MW);
    dim3 grid((unsigned)nmsgtiles, (args.rb + WN - 1) / WN), block(QF_NW * 64);
    QfGemmArgs la = args;
    la.mchunk = qf_gemm_mchunk(BITS, nmsgtiles, grid.y);
    QfTimer t(ctx, &ctx->ms_gemm);
    hipLaunchKernelGGL((qf_gemm_kernel<LA, LB, WN, MW, BITS>), grid, block, lds, ctx->stream, la);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

template <int LA, int WN>
static int qf_gemm_dispatch_packed(qf_ctx *ctx, const QfGemmArgs &args) {
    size_t lds = 3 * (size_t)WN * LA * QF_KS * 1024;
    static QfPerDevice attr;
    QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_gemm_packed_kernel<LA, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
    uint64_t nmsgtiles = (args.B + 32 * QF_NW - 1) / (32 * QF_NW);
    dim3 grid((unsigned)nmsgtiles, args.rb / WN), block(QF_NW * 64);   // rb = full blocks + 1, a multiple of WN here
    QfGemmArgs la = args;
    la.mchunk = qf_gemm_mchunk(true, nmsgtiles, grid.y);
    QfTimer t(ctx, &ctx->ms_gemm);
    hipLaunchKernelGGL((qf_gemm_packed_kernel<LA, WN>), grid, block, lds, ctx->stream, la);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

template <int LA, int WN>
static int qf_gemm_dispatch_packedx(qf_ctx *ctx, const QfGemmArgs &args) {
    size_t lds = 3 * ((size_t)WN * LA + 1) * QF_KS * 1024;
    static QfPerDevice attr;
    QF_HIP(ctx, attr.once(ctx->device, [&] { return hipFuncSetAttribute((const void *)qf_gemm_packedx_kernel<LA, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }));
    uint64_t nmsgtiles = (args.B + 32 * QF_NW - 1) / (32 * QF_NW);
    dim3 grid((unsigned)nmsgtiles, (args.rb - 1) / WN), block(QF_NW * 64);   // rb - 1 full blocks, a multiple of WN here
    QfGemmArgs la = args;
    la.mchunk = qf_gemm_mchunk(true, nmsgtiles, grid.y);
    QfTimer t(ctx, &ctx->ms_gemm);
    hipLaunchKernelGGL((qf_gemm_packedx_kernel<LA, WN>), grid, block, lds, ctx->stream, la);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// Wave tile = MW*32 messages x WN*32 key rows; its accumulators (16 registers per 32x32 block and
// digit weight) must leave room for operand fragments inside the 256-VGPR budget of 2 waves/SIMD.
template <int LA, int LB, bool BITS>
static int qf_gemm_dispatch(qf_ctx *ctx, const QfGemmArgs &args) {
    constexpr int S = LA + LB - 1;
    constexpr int ATREGS = BITS ? 0 : (QF_KS * LB <= 8 ? 8 * QF_KS * LB : 0);
    constexpr int BUDGET = (224 - ATREGS) < 192 ? (224 - ATREGS) : 192;
    constexpr int WFIT = BUDGET / (S * 16);
    constexpr int WMAX = WFIT < 1 ? 1 : (WFIT > 4 ? 4 : WFIT);
    // pick the tile width by cost = padded row blocks x (1 + 0.3/w): a wider tile re-uses each
    // batch fragment for more MFMAs (measured: LPR n=1024 is 30% faster at w=2 than at w=1 although
    // 33 row blocks pad to 34), a narrower one wastes fewer padded rows (Regev: 9 blocks -> w=3)
    int best = 1;
    double best_cost = 1e30;
    for (int w = WMAX; w >= 1; w--) {
        uint32_t tiles = (args.rb + w - 1) / w;
        double cost = (double)tiles * w * (1.0 + 0.3 / w);
        if (cost < best_cost) { best_cost = cost; best = w; }
    }
    if constexpr (BITS) {
        // packed remainder block: needs the slots (full blocks + 1) to split evenly into tiles of w
        if (args.rem_planes && !getenv("QF_GEMM_NOPACK")) {
            int bw = 0;
            double bc = best_cost;
            for (int w = WMAX; w >= 1; w--) {
                if (args.rb % w) continue;
                double cost = ((double)args.rb * LA - (LA - 1)) / LA * (1.0 + 0.3 / w);
                if (cost < bc) { bc = cost; bw = w; }
            }
            // ... or as an extra fragment of the last tile when the FULL blocks split evenly into tiles of
            // w and (w*LA + 1) accumulator sets + 3 x (w*LA + 1) KiB-fragments x KS of LDS still fit
            if constexpr (LA <= 3) if (!getenv("QF_GEMM_NOPACKX")) {   // LA = 4: 13 accumulator sets + operands do not fit
                int xw = 0;
                for (int w = WMAX; w >= 2; w--) {
                    if (args.rb < 2 || (args.rb - 1) % w) continue;
#ifndef QF_GEMM_MFMA32
                    if (args.rem_rows * LA > 16) continue;   // the 16x16x64 body keeps only the first row half of the extra fragment
#endif
                    if ((w * LA + 1) * 16 > 208 || 3 * (w * LA + 1) * QF_KS > 160) continue;
                    double cost = ((double)(args.rb - 1) * LA + 1) / LA * (1.0 + 0.3 / w);
                    if (cost < bc) { bc = cost; xw = w; bw = 0; }
                }
                switch (xw) {
                    case 2: if constexpr (WMAX >= 2) return qf_gemm_dispatch_packedx<LA, 2>(ctx, args); break;
                    case 3: if constexpr (WMAX >= 3) return qf_gemm_dispatch_packedx<LA, 3>(ctx, args); break;
                    case 4: if constexpr (WMAX >= 4) return qf_gemm_dispatch_packedx<LA, 4>(ctx, args); break;
                }
            }
            switch (bw) {
                case 1: return qf_gemm_dispatch_packed<LA, 1>(ctx, args);
                case 2: if constexpr (WMAX >= 2) return qf_gemm_dispatch_packed<LA, 2>(ctx, args); break;
                case 3: if constexpr (WMAX >= 3) return qf_gemm_dispatch_packed<LA, 3>(ctx, args); break;
                case 4: if constexpr (WMAX >= 4) return qf_gemm_dispatch_packed<LA, 4>(ctx, args); break;
            }
        }
    }
    int mw = 1;
    if (const char *e = getenv("QF_GEMM_WN")) {  // experiments only
        int w = atoi(e);
        if (w >= 1 && w <= WMAX) best = w;
    }
    if (const char *e = getenv("QF_GEMM_MW")) mw = atoi(e);
    if (args.xkt) {   // k-tile-major bits: the 16x16x64 body only (every default bit-operand path)
#ifdef QF_GEMM_MFMA32
        return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: k-tile-major bits need the 16x16x64 body");
#endif
        mw = 1;
    }
    constexpr bool M2W2 = 2 * 2 * S * 16 <= BUDGET, M2W1 = 2 * 1 * S * 16 <= BUDGET;
    if (mw == 2) {
        if (best >= 2) { if constexpr (M2W2) return qf_gemm_dispatch_wn<LA, LB, 2, 2, BITS>(ctx, args); }
        if constexpr (M2W1) return qf_gemm_dispatch_wn<LA, LB, 1, 2, BITS>(ctx, args);
    }
    switch (best) {
        case 1: return qf_gemm_dispatch_wn<LA, LB, 1, 1, BITS>(ctx, args);
        case 2: if constexpr (WMAX >= 2) return qf_gemm_dispatch_wn<LA, LB, 2, 1, BITS>(ctx, args); break;
        case 3: if constexpr (WMAX >= 3) return qf_gemm_dispatch_wn<LA, LB, 3, 1, BITS>(ctx, args); break;
        case 4: if constexpr (WMAX >= 4) return qf_gemm_dispatch_wn<LA, LB, 4, 1, BITS>(ctx, args); break;
    }
    return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: no tile configuration");
}

// builds key->fold_planes[lb] on first use (see qf_key)
static int qf_key_fold(qf_ctx *ctx, const qf_key *key, int lb) {
    if (lb < 2 || lb > 4) return qf_fail(ctx, QF_ERR_ARG, "fold: operand digits must be 2..4");
    // two contexts (threads) of one device may meet here with the same key: one builds, the other waits
    static std::mutex fold_mu;
    std::lock_guard<std::mutex> lk(fold_mu);
    if (key->fold_planes[lb]) return QF_OK;
    int8_t *planes = nullptr;
    size_t bytes = (size_t)key->rb * key->ks * lb * key->la * 1024;
    if (hipMalloc((void **)&planes, bytes) != hipSuccess) return qf_fail(ctx, QF_ERR_DEVICE, "hipMalloc(folded key planes) failed");
    uint64_t mult = 1 % key->q;
    for (int b = 0; b < lb; b++) {
        int rc = qf_launch_recode(ctx, key->u32, QF_X_U32, (size_t)key->cols * 4, key->rows, key->cols, key->mod, key->la,
                                  key->ks, planes, ctx->flag, (uint32_t)mult, (uint32_t)lb, (uint32_t)b);
        if (rc) { hipFree(planes); return rc; }
        mult = (mult * 256) % key->q;
    }
    // the planes must be complete before another context (another stream) can find them
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { hipFree(planes); return qf_fail(ctx, QF_ERR_DEVICE, "folded key planes: recode failed"); }
    key->fold_planes[lb] = planes;
    return QF_OK;
}

int qf_launch_gemm(qf_ctx *ctx, const qf_key *key, const void *X, int x_is_bits, size_t x_stride,
                   int lb, uint64_t B, const uint32_t *addend, const uint8_t *msg, uint32_t *out) {
    QfGemmArgs a;
    a.kplanes = key->planes; a.rb = key->rb; a.ks = key->ks; a.rows = key->rows;
    a.rem_planes = (x_is_bits && lb == 1) ? key->rem_planes : nullptr; a.rem_rows = key->rem_rows;
    a.xbits = x_is_bits ? (const uint8_t *)X : nullptr;
    a.xkt = x_is_bits == 2;   // k-tile-major
    a.xstride = x_stride;
    a.xplanes = x_is_bits ? nullptr : (const int8_t *)X;
    a.mchunk = 0;
    a.B = B; a.addend = addend; a.msg = msg; a.out = out; a.mod = key->mod; a.dbg = g_qf_dbg;
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
    a.xlb_total = (uint32_t)lb; a.xlb_off = 0;
    if (key->rows >= (1u << 24)) return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: more than 2^24 key rows (36 output rows are addressed through a 32-bit offset)");
    // fp64 epilogue of the one-digit kernels (qf_epi_f64): |acc| <= K * (bits: 2^7, digits: 2^14), K counted
    // with the folded digits; recombination in one step while 257 * 65536 * |acc| + 2^34 < 2^52
    a.qd = (double)key->q; a.qinv = 1.0 / (double)key->q;
    {
        long double accmax = (long double)key->ks * 32.0L * (x_is_bits ? 128.0L : 16384.0L * lb);
        a.epi2 = accmax * 257.0L * 65536.0L + 17179869184.0L >= 4503599627370496.0L ? 1u : 0u;
        if (const char *e = getenv("QF_GEMM_EPI2")) a.epi2 = atoi(e) != 0;   // tests: force the two-step recombination
    }
    if (x_is_bits && lb != 1) return qf_fail(ctx, QF_ERR_ARG, "gemm: bit operand has one digit");
    // Operand digits folded into the inner dimension (default): A*x = [A | 256A | ...]*[x_0; x_1; ...]
    // over LB*K columns with the single-digit kernel.  Same MFMA count as the fused <LA,LB> kernel,
    // S = LA accumulators per row block, ONE pass (no output re-read), the batch planes are consumed
    // exactly as the recode kernel lays them out ([mb][ks][lb] = k-step ks*LB + b).  Needs
    // LB*K*2^14 < 2^31 for the int32 accumulators -- which the check above already demands of every
    // product with LB <= LA, so the fused <LA,LB> kernels below only run under QF_GEMM_FOLD=0 (tests).
    bool fold = !x_is_bits && lb > 1 && (long double)lb * key->ks * 32.0L * 16384.0L < 2147483648.0L;
    if (const char *e = getenv("QF_GEMM_FOLD")) fold = fold && atoi(e) != 0;
    if (fold) {
        int rc = qf_key_fold(ctx, key, lb);
        if (rc) return rc;
        QfGemmArgs p = a;
        p.kplanes = key->fold_planes[lb];
        p.ks = key->ks * (uint32_t)lb;
        p.xlb_total = 1; p.xlb_off = 0;
        switch (qf_gemm_shortk_wn(p, la) * 8 + la) {
            case 2 * 8 + 1: return qf_gemm_dispatch_shortk<1, 2, 1>(ctx, p);
            case 2 * 8 + 2: return qf_gemm_dispatch_shortk<2, 2, 1>(ctx, p);
            case 2 * 8 + 3: return qf_gemm_dispatch_shortk<3, 2, 1>(ctx, p);
            case 2 * 8 + 4: return qf_gemm_dispatch_shortk<4, 2, 1>(ctx, p);
            case 1 * 8 + 1: return qf_gemm_dispatch_shortk<1, 1, 2>(ctx, p);
            case 1 * 8 + 2: return qf_gemm_dispatch_shortk<2, 1, 2>(ctx, p);
            case 1 * 8 + 3: return qf_gemm_dispatch_shortk<3, 1, 2>(ctx, p);
        }
        switch (la) {
            case 1: return qf_gemm_dispatch<1, 1, false>(ctx, p);
            case 2: return qf_gemm_dispatch<2, 1, false>(ctx, p);
            case 3: return qf_gemm_dispatch<3, 1, false>(ctx, p);
            case 4: return qf_gemm_dispatch<4, 1, false>(ctx, p);
        }
        return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: digit count not instantiated");
    }
#define QF_CASE(LA_, LB_)                                                            \
    if (la == LA_ && lb == LB_) {                                                    \
        return x_is_bits ? qf_gemm_dispatch<LA_, 1, true>(ctx, a)                    \
                         : qf_gemm_dispatch<LA_, LB_, false>(ctx, a);                \
    }
    QF_CASE(1, 1) QF_CASE(2, 1) QF_CASE(3, 1) QF_CASE(4, 1)
    QF_CASE(2, 2) QF_CASE(3, 2) QF_CASE(4, 2)
    QF_CASE(3, 3) QF_CASE(4, 3)
    QF_CASE(4, 4)
#undef QF_CASE
    return qf_fail(ctx, QF_ERR_NOTSUP, "gemm: digit combination not instantiated (need LB <= LA <= 4)");
}
