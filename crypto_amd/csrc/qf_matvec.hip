// Kernel (c): HBM-streaming single-message products and the batched dot-product decryption.
// These are byte-streaming kernels: every matrix / ciphertext word is read exactly once with
// 16-byte-per-lane coalesced loads, products are accumulated exactly as split 64-bit sums
// (low and high 32-bit halves of each 62-bit product are summed separately, so nothing can
// overflow), and the reduction mod q happens once per output.
#include "qf_internal.h"

#define MV_BLOCK 256
#define MV_CHUNK 4096  // columns handled by one workgroup of the mat-vec (16 per thread)

struct Acc2 { uint64_t lo, hi; };
__device__ __forceinline__ void acc_mad(Acc2 &a, uint32_t x, uint32_t y) {
    uint64_t p = (uint64_t)x * y;
    a.lo += (uint32_t)p;
    a.hi += p >> 32;
}
__device__ __forceinline__ uint32_t acc_finish(uint64_t lo, uint64_t hi, const QfMod &m) {
    // value = hi*2^32 + lo  (hi, lo < 2^64 each)
    uint32_t h = qf_red_u64(hi, m), l = qf_red_u64(lo, m);
    return qf_addmod(qf_mulmod(h, m.r32, m), l, m.q);
}
__device__ __forceinline__ Acc2 block_sum(Acc2 a, Acc2 *sm) {
    a.lo = qf_wave_sum_u64(a.lo);
    a.hi = qf_wave_sum_u64(a.hi);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = a;
    __syncthreads();
    Acc2 t = {0, 0};
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) { t.lo += sm[i].lo; t.hi += sm[i].hi; }
    return t;
}

// y = A*x : grid (chunks, rows); partial[row][chunk] = split sum over the chunk's columns
__global__ __launch_bounds__(MV_BLOCK) void qf_matvec_partial(const uint32_t *__restrict__ A,
                                                              uint32_t rows, uint32_t cols,
                                                              const uint32_t *__restrict__ x,
                                                              Acc2 *__restrict__ partial, uint32_t nchunk) {
    __shared__ Acc2 sm[MV_BLOCK / 64];
    const uint32_t row = blockIdx.y, chunk = blockIdx.x;
    const uint32_t c0 = chunk * MV_CHUNK;
    const uint32_t *a = A + (size_t)row * cols;
    Acc2 acc = {0, 0};
    const bool vec_ok = ((((size_t)row * cols) & 3) == 0) && ((((uintptr_t)A) & 15) == 0) && ((((uintptr_t)x) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MV_CHUNK / (MV_BLOCK * 4); i++) {
        uint32_t c = c0 + (i * MV_BLOCK + threadIdx.x) * 4;
        if (vec_ok && c + 4 <= cols) {
            uint4 av = *(const uint4 *)(a + c), xv = *(const uint4 *)(x + c);
            acc_mad(acc, av.x, xv.x); acc_mad(acc, av.y, xv.y);
            acc_mad(acc, av.z, xv.z); acc_mad(acc, av.w, xv.w);
        } else {
            for (uint32_t k = c; k < c + 4 && k < cols; k++) acc_mad(acc, a[k], x[k]);
        }
    }
    Acc2 t = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[(size_t)row * nchunk + chunk] = t;
}
__global__ void qf_matvec_finish(const Acc2 *__restrict__ partial, uint32_t n, uint32_t nchunk,
                                 QfMod mod, uint32_t *__restrict__ y) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = 0, hi = 0;
    for (uint32_t c = 0; c < nchunk; c++) { lo += partial[(size_t)i * nchunk + c].lo; hi += partial[(size_t)i * nchunk + c].hi; }
    y[i] = acc_finish(lo, hi, mod);
}

// y = A*x in ONE launch when every row starts 16-byte aligned: a workgroup streams a whole row
// (all its loads are independent and issued back to back), reduces it and writes the canonical
// residue -- no partial-sum buffer, no second kernel.
__global__ __launch_bounds__(MV_BLOCK) void qf_matvec_rows(const uint32_t *__restrict__ A, uint32_t cols,
                                                           const uint32_t *__restrict__ x, QfMod mod,
                                                           uint32_t *__restrict__ y) {
    __shared__ Acc2 sm[MV_BLOCK / 64];
    const uint32_t row = blockIdx.x;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *a = (const u32x4 *)(A + (size_t)row * cols);
    const u32x4 *xv4 = (const u32x4 *)x;
    const uint32_t nv = cols >> 2;  // cols % 4 == 0 (checked by the launcher)
    Acc2 acc = {0, 0};
#pragma unroll 4
    for (uint32_t i = threadIdx.x; i < nv; i += MV_BLOCK) {
        const u32x4 av = __builtin_nontemporal_load(a + i), xv = xv4[i];
        acc_mad(acc, av.x, xv.x); acc_mad(acc, av.y, xv.y);
        acc_mad(acc, av.z, xv.z); acc_mad(acc, av.w, xv.w);
    }
    Acc2 t = block_sum(acc, sm);
    if (threadIdx.x == 0) y[row] = acc_finish(t.lo, t.hi, mod);
}

// y^t = x^t*A : grid (column tiles of 1024, row chunks); thread owns 4 adjacent columns
#define VM_ROWS 64
__global__ __launch_bounds__(MV_BLOCK) void qf_vecmat_partial(const uint32_t *__restrict__ A,
                                                              uint32_t rows, uint32_t cols,
                                                              const uint32_t *__restrict__ x,
                                                              Acc2 *__restrict__ partial /* [rchunk][cols] */) {
    const uint32_t c = (blockIdx.x * MV_BLOCK + threadIdx.x) * 4;
    const uint32_t r0 = blockIdx.y * VM_ROWS;
    const uint32_t r1 = min(rows, r0 + VM_ROWS);
    Acc2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const bool vec_ok = ((cols & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
    if (c < cols) {
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t xr = x[r];
            const uint32_t *a = A + (size_t)r * cols + c;
            if (vec_ok && c + 4 <= cols) {
                uint4 av = *(const uint4 *)a;
                acc_mad(acc[0], av.x, xr); acc_mad(acc[1], av.y, xr);
                acc_mad(acc[2], av.z, xr); acc_mad(acc[3], av.w, xr);
            } else {
                for (uint32_t k = 0; k < 4 && c + k < cols; k++) acc_mad(acc[k], a[k], xr);
            }
        }
        for (uint32_t k = 0; k < 4 && c + k < cols; k++) partial[(size_t)blockIdx.y * cols + c + k] = acc[k];
    }
}
__global__ void qf_vecmat_finish(const Acc2 *__restrict__ partial, uint32_t cols, uint32_t nchunk,
                                 QfMod mod, uint32_t *__restrict__ y) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    uint64_t lo = 0, hi = 0;
    for (uint32_t c = 0; c < nchunk; c++) { lo += partial[(size_t)c * cols + j].lo; hi += partial[(size_t)c * cols + j].hi; }
    y[j] = acc_finish(lo, hi, mod);
}

int qf_launch_matvec(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y) {
    if (key->rows >= 128 && (key->cols & 3) == 0 && (((uintptr_t)key->u32) & 15) == 0 && (((uintptr_t)x) & 15) == 0) {
        QfTimer t(ctx, &ctx->ms_matvec);
        hipLaunchKernelGGL(qf_matvec_rows, dim3(key->rows), dim3(MV_BLOCK), 0, ctx->stream, key->u32, key->cols, x,
                           key->mod, y);
        QF_HIP(ctx, hipGetLastError());
        return QF_OK;
    }
    uint32_t nchunk = (key->cols + MV_CHUNK - 1) / MV_CHUNK;
    int rc = qf_ws2_reserve(ctx, (size_t)key->rows * nchunk * sizeof(Acc2));
    if (rc) return rc;
    dim3 grid(nchunk, key->rows), block(MV_BLOCK);
    {
        QfTimer t(ctx, &ctx->ms_matvec);
        hipLaunchKernelGGL(qf_matvec_partial, grid, block, 0, ctx->stream, key->u32, key->rows, key->cols, x,
                           (Acc2 *)ctx->ws2, nchunk);
    }
    hipLaunchKernelGGL(qf_matvec_finish, dim3((key->rows + 255) / 256), dim3(256), 0, ctx->stream,
                       (const Acc2 *)ctx->ws2, key->rows, nchunk, key->mod, y);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_vecmat(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y) {
    uint32_t nchunk = (key->rows + VM_ROWS - 1) / VM_ROWS;
    int rc = qf_ws2_reserve(ctx, (size_t)key->cols * nchunk * sizeof(Acc2));
    if (rc) return rc;
    dim3 grid((key->cols + MV_BLOCK * 4 - 1) / (MV_BLOCK * 4), nchunk), block(MV_BLOCK);
    {
        QfTimer t(ctx, &ctx->ms_matvec);
        hipLaunchKernelGGL(qf_vecmat_partial, grid, block, 0, ctx->stream, key->u32, key->rows, key->cols, x,
                           (Acc2 *)ctx->ws2);
    }
    hipLaunchKernelGGL(qf_vecmat_finish, dim3((key->cols + 255) / 256), dim3(256), 0, ctx->stream,
                       (const Acc2 *)ctx->ws2, key->cols, nchunk, key->mod, y);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// Batched decryption: a wave handles DEC_U ciphertexts per iteration (all their loads in flight
// together), lanes stride over the n entries.  Each lane folds its split 64-bit sums to a canonical
// residue once; the wave sum is then taken mod q with DPP adds inside each row of 16 lanes (no LDS
// traffic) and four scalar reads.
#define DEC_U 8
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_addmod(uint32_t v, uint32_t q) {
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
    uint32_t s = v + o;  // both < q < 2^31
    return min(s, s - q);
}
__device__ __forceinline__ uint32_t wave_sum_modq(uint32_t v, uint32_t q) {
    v = dpp_addmod<0xB1>(v, q);   // quad_perm [1,0,3,2]
    v = dpp_addmod<0x4E>(v, q);   // quad_perm [2,3,0,1]
    v = dpp_addmod<0x141>(v, q);  // row_half_mirror: the other quad of each 8
    v = dpp_addmod<0x140>(v, q);  // row_mirror: the other half of each row of 16
    uint32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    uint32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    uint32_t a = r0 + r1; a = min(a, a - q);
    uint32_t b = r2 + r3; b = min(b, b - q);
    a += b;
    return min(a, a - q);
}
__global__ __launch_bounds__(256) void qf_dot_decode_kernel(const uint32_t *__restrict__ w, uint32_t n,
                                                            QfMod mod, const uint32_t *__restrict__ C,
                                                            uint64_t B, uint8_t *__restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t stride = n + 1;
    for (uint64_t b0 = wave * DEC_U; b0 < B; b0 += nwaves * DEC_U) {
        const uint32_t *c[DEC_U];
        Acc2 acc[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            const uint64_t b = b0 + u < B ? b0 + u : B - 1;  // clamped: the duplicate result is dropped
            c[u] = C + b * stride;
            acc[u].lo = 0; acc[u].hi = 0;
        }
        uint32_t last[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; u++) last[u] = lane == 0 ? c[u][n] : 0u;
#pragma unroll 4
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t wv = w[i];
#pragma unroll
            for (int u = 0; u < DEC_U; u++) acc_mad(acc[u], wv, c[u][i]);
        }
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            uint32_t r = acc_finish(acc[u].lo, acc[u].hi, mod);
            r = qf_addmod(r, qf_red_u64(last[u], mod), mod.q);
            r = wave_sum_modq(r, mod.q);
            if (lane == 0 && b0 + u < B) bits[b0 + u] = (uint8_t)qf_decide_bit(r, mod);
        }
    }
}
int qf_launch_dot_decode(qf_ctx *ctx, const uint32_t *w, uint32_t n, const QfMod &mod, const uint32_t *C,
                         uint64_t B, uint8_t *bits) {
    uint64_t blocks = (B + 4 * DEC_U - 1) / (4 * DEC_U);  // 4 waves per block, DEC_U ciphertexts per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) return QF_OK;
    QfTimer t(ctx, &ctx->ms_decode);
    hipLaunchKernelGGL(qf_dot_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w, n, mod, C, B, bits);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
