// Counter-mode generator on the device (include/qfgpu.h "seeded generator"): SplitMix64 under a 64-bit
// seed or ChaCha20 under a 256-bit key (QfGen, qf_device.h).
// Bit-identical to oracle/qf_oracle.c: qfo_sample_uniform / qfo_sample_bits / qfo_sample_noise.
#include "qf_internal.h"

__global__ void qf_sample_uniform_kernel(QfGen base, uint64_t first, uint64_t count, uint32_t q,
                                         uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) out[i] = qf_gen_below(base, first + i, q);
}

// Bit rows: row r takes W64 = ceil(cols / 64) generator values; bit j of the row is bit (j & 63) of
// value((first_row + r) * W64 + (j >> 6)) -- all 64 bits of a draw are used (one SplitMix64 per bit made
// this kernel VALU-bound: 174 us for 65,536 x 6,168 bits, 60 % of the GEMM that consumes them; now it
// writes at HBM speed).  One thread per output dword; stride_bytes is a multiple of 4.
__global__ void qf_sample_bits_kernel(QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols,
                                      uint32_t stride_bytes, uint8_t *__restrict__ out) {
    const uint32_t wpr = stride_bytes / 4, w64 = (cols + 63) / 64;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = rows * wpr, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const uint64_t r = i / wpr;
        const uint32_t d = (uint32_t)(i % wpr), j0 = d * 32;      // this dword holds bits j0 .. j0+31 of the row
        uint32_t word = 0;
        if (j0 < cols) {
            const uint64_t v = qf_gen64(base, (first_row + r) * w64 + (d >> 1));
            word = (uint32_t)(v >> (32 * (d & 1)));
            if (cols - j0 < 32) word &= (1u << (cols - j0)) - 1u;   // padding bits are zero
        }
        ((uint32_t *)out)[i] = word;
    }
}

// The same rows, 16 bytes (two draws) per thread: rows whose stride is a multiple of 16 bytes in a
// 16-byte aligned buffer (what the GEMM's bit operand requires anyway).  No 64-bit division, each draw
// computed once, 16-byte stores: grid.x * 256 threads cover the 16-byte groups of a row, grid.y strides rows.
__global__ __launch_bounds__(256) void qf_sample_bits16_kernel(QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols,
                                                               uint32_t stride_bytes, uint8_t *__restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;       // 16-byte group of the row: bits 128g .. 128g+127
    if (g >= stride_bytes / 16) return;
    const uint32_t w64 = (cols + 63) / 64;
    for (uint64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        uint64_t v[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t j0 = (2 * g + t) * 64;                   // first bit of this draw
            v[t] = 0;
            if (j0 < cols) {
                v[t] = qf_gen64(base, (first_row + r) * w64 + 2 * g + t);
                if (cols - j0 < 64) v[t] &= (1ull << (cols - j0)) - 1ull;   // padding bits are zero
            }
        }
        uint4 o;
        o.x = (uint32_t)v[0]; o.y = (uint32_t)(v[0] >> 32); o.z = (uint32_t)v[1]; o.w = (uint32_t)(v[1] >> 32);
        *(uint4 *)(out + r * stride_bytes + (size_t)g * 16) = o;
    }
}

// The same bits written k-tile-major (QF_X_BITS_KT): the 16-byte word of row r, k-tile g goes to g * kt_stride + 16 r.
// A thread per row (consecutive rows = consecutive words), blockIdx.y = k-tile; the draws are those of the row-major kernel.
__global__ __launch_bounds__(256) void qf_sample_bits_kt_kernel(QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols,
                                                                size_t kt_stride, uint8_t *__restrict__ out) {
    const uint32_t g = blockIdx.y;
    const uint32_t w64 = (cols + 63) / 64;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t j0 = (2 * g + t) * 64;
            v[t] = 0;
            if (j0 < cols) {
                v[t] = qf_gen64(base, (first_row + r) * w64 + 2 * g + t);
                if (cols - j0 < 64) v[t] &= (1ull << (cols - j0)) - 1ull;
            }
        }
        uint4 o;
        o.x = (uint32_t)v[0]; o.y = (uint32_t)(v[0] >> 32); o.z = (uint32_t)v[1]; o.w = (uint32_t)(v[1] >> 32);
        *(uint4 *)(out + (size_t)g * kt_stride + r * 16) = o;
    }
}

__global__ void qf_sample_noise_kernel(QfGen base, uint64_t first, uint64_t count, QfMod mod, uint32_t W,
                                       uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        int64_t v = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) v += (int64_t)qf_gen_below(base, (first + i) * 4 + t, 2 * W + 1) - (int64_t)W;
        out[i] = qf_red_i64(v, mod);
    }
}

// Discrete Gaussian D_{Z,s} (center 0) by inversion: thr[i] = floor(2^64 * P[X <= lo + i]) built on the
// host (qf_api.hip, build_cdt), one 64-bit draw per sample, x = lo + #{i : thr[i] <= v}.
__global__ void qf_sample_cdt_kernel(QfGen base, uint64_t first, uint64_t count, QfMod mod,
                                     const uint64_t *__restrict__ thr, uint32_t len, int64_t lo,
                                     uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const uint64_t v = qf_gen64(base, first + i);
        uint32_t a = 0, b = len - 1;          // smallest index with thr[idx] > v; thr[len-1] = 2^64-1 ends the search
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if (thr[m] > v) b = m; else a = m + 1;
        }
        out[i] = qf_red_i64(lo + (int64_t)a, mod);
    }
}

static unsigned sampler_grid(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b ? b : 1));
}
int qf_launch_sample_uniform(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, uint32_t q, uint32_t *out) {
    if (!count) return QF_OK;
    hipLaunchKernelGGL(qf_sample_uniform_kernel, dim3(sampler_grid(count)), dim3(256), 0, ctx->stream, base, first, count, q, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_sample_bits(qf_ctx *ctx, QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols,
                          uint32_t stride_bytes, uint8_t *out) {
    if (!rows) return QF_OK;
    if (stride_bytes % 16 == 0 && ((uintptr_t)out & 15) == 0) {
        const unsigned gx = (stride_bytes / 16 + 255) / 256;
        uint64_t gy = rows < 65535 ? rows : 65535;
        while (gy > 1 && (uint64_t)gx * gy > 256 * 64) gy = (gy + 1) / 2;     // enough waves to fill the chip, rows strided
        hipLaunchKernelGGL(qf_sample_bits16_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, ctx->stream,
                           base, first_row, rows, cols, stride_bytes, out);
    } else
    hipLaunchKernelGGL(qf_sample_bits_kernel, dim3(sampler_grid(rows * (stride_bytes / 4))), dim3(256), 0, ctx->stream,
                       base, first_row, rows, cols, stride_bytes, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_sample_bits_kt(qf_ctx *ctx, QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols, size_t kt_stride, uint8_t *out) {
    if (!rows || !cols) return QF_OK;
    const uint32_t ntiles = (cols + 127) / 128;
    uint64_t gx = (rows + 255) / 256;
    while (gx > 1 && gx * ntiles > 256 * 64) gx = (gx + 1) / 2;   // enough waves to fill the chip, rows strided
    hipLaunchKernelGGL(qf_sample_bits_kt_kernel, dim3((unsigned)gx, ntiles), dim3(256), 0, ctx->stream, base, first_row, rows, cols, kt_stride, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_sample_noise(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, const QfMod &mod, uint32_t W, uint32_t *out) {
    if (!count) return QF_OK;
    hipLaunchKernelGGL(qf_sample_noise_kernel, dim3(sampler_grid(count)), dim3(256), 0, ctx->stream, base, first, count, mod, W, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_sample_cdt(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, const QfMod &mod,
                         const uint64_t *thr_dev, uint32_t len, int64_t lo, uint32_t *out) {
    if (!count) return QF_OK;
    hipLaunchKernelGGL(qf_sample_cdt_kernel, dim3(sampler_grid(count)), dim3(256), 0, ctx->stream, base, first, count, mod, thr_dev, len, lo, out);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
