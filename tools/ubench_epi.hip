// Micro-benchmark + equivalence check of two GEMM-epilogue reductions on gfx950 (not part of the
// product): (A) int64 recombination with centred constants + 64-bit Barrett (qf_red_i64),
// (B) exact fp64 recombination V = sum acc[s]*2^(8s) (+ addend), quotient by fp64 multiply + floor,
// remainder by one exact fma, two fix-ups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../crypto_amd/csrc/qf_device.h"
#define ITER 2048
static QfMod make_mod(uint32_t q) {
    QfMod m; m.q = q; m.qhalf = q / 2;
    m.mu = (uint64_t)(((unsigned __int128)1 << 64) / q);
    uint64_t t = (1ull << 62) / q * q; if (t < (1ull << 62)) t += q; m.off = t;
    m.r32 = (uint32_t)((1ull << 32) % q); m.pad = 0; return m;
}
template <int S>
__device__ __forceinline__ uint32_t red_a(const int32_t *acc, const int32_t *cs, uint32_t addend, const QfMod &m) {
    int64_t x = 0;
#pragma unroll
    for (int s = 0; s < S; s++) x += (int64_t)acc[s] * (int64_t)cs[s];
    x += (int64_t)addend;
    return qf_red_i64(x, m);
}
template <int S>
__device__ __forceinline__ uint32_t red_b(const int32_t *acc, uint32_t addend, double qd, double qinv, uint32_t q) {
    double v = (double)addend;
    double sc = 1.0;
#pragma unroll
    for (int s = 0; s < S; s++) { v = fma((double)acc[s], sc, v); sc *= 256.0; }
    double qf = floor(v * qinv);
    int32_t r = (int32_t)fma(-qf, qd, v);
    r += (r < 0) ? (int32_t)q : 0;
    r -= (r >= (int32_t)q) ? (int32_t)q : 0;
    return (uint32_t)r;
}
template <int S, int WHICH>
__global__ __launch_bounds__(256) void k(uint32_t *out, QfMod m, int32_t c0, int32_t c1, int32_t c2, int32_t c3, double qd, double qinv, int32_t amax) {
    const int32_t cs[4] = {c0, c1, c2, c3};
    uint32_t st = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t sum = 0, mism = 0;
    for (int i = 0; i < ITER; i++) {
        int32_t acc[4];
#pragma unroll
        for (int s = 0; s < 4; s++) { st = st * 1664525u + 1013904223u; acc[s] = (int32_t)(st % (2u * amax + 1)) - amax; }
        st = st * 1664525u + 1013904223u;
        uint32_t ad = st % m.q;
        if (WHICH == 0) sum += red_a<S>(acc, cs, ad, m);
        if (WHICH == 1) sum += red_b<S>(acc, ad, qd, qinv, m.q);
        if (WHICH == 2) mism += red_a<S>(acc, cs, ad, m) != red_b<S>(acc, ad, qd, qinv, m.q);
        if (WHICH == 3) sum += acc[0] + acc[1] + acc[2] + acc[3] + ad;   // generator cost alone
    }
    out[blockIdx.x * 256 + threadIdx.x] = WHICH == 2 ? mism : sum;
}
int main() {
    uint32_t *out; hipMalloc(&out, 2048 * 256 * 4);
    uint32_t *h = (uint32_t *)malloc(2048 * 256 * 4);
    struct { uint32_t q; int S; int32_t amax; } cases[] = {{16777213u, 3, 6168 * 128}, {1073741789u, 4, 2048 * 128 * 128}, {8380417u, 3, 11776 * 128},
                                                           {7u, 1, 100}, {16u, 2, 1 << 30}, {2147483629u, 4, 16384 * 16384 - 1}};
    for (auto &c : cases) {
        QfMod m = make_mod(c.q);
        int32_t cs[4]; uint64_t p = 1 % c.q;
        for (int s = 0; s < 4; s++) { int64_t v = (int64_t)p; if (v > (int64_t)(c.q / 2)) v -= c.q; cs[s] = (int32_t)v; p = p * 256 % c.q; }
        double qd = c.q, qinv = 1.0 / qd;
        float ms[4];
        for (int w = 0; w < 4; w++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
#define L(S_, W_) hipLaunchKernelGGL((k<S_, W_>), dim3(2048), dim3(256), 0, 0, out, m, cs[0], cs[1], cs[2], cs[3], qd, qinv, c.amax)
#define LS(W_) switch (c.S) { case 1: L(1, W_); break; case 2: L(2, W_); break; case 3: L(3, W_); break; default: L(4, W_); }
                switch (w) { case 0: LS(0) break; case 1: LS(1) break; case 2: LS(2) break; default: LS(3) break; }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms[w], e0, e1);
            if (w == 2) {
                hipMemcpy(h, out, 2048 * 256 * 4, hipMemcpyDeviceToHost);
                uint64_t mm = 0; for (int i = 0; i < 2048 * 256; i++) mm += h[i];
                printf("q=%u S=%d amax=%d mismatches=%llu of %llu\n", c.q, c.S, c.amax, (unsigned long long)mm, (unsigned long long)2048 * 256 * ITER);
            }
        }
        double n = 2048.0 * 256 * ITER;
        printf("   int64+Barrett %.3f ms, fp64 %.3f ms, generator %.3f ms -> net %.2f vs %.2f T reductions/s\n", ms[0], ms[1], ms[3],
               n / (ms[0] - ms[3]) * 1e-9, n / (ms[1] - ms[3]) * 1e-9);
    }
    return 0;
}
