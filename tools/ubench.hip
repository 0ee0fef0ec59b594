// Micro-benchmarks that decide kernel design on gfx950: integer multiply rates (NTT butterflies),
// fp64 fma rate, int8 MFMA issue rate.  Prints ops/clk/CU-equivalents.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define N_ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b + 12345u, d = a + 77u;
    uint32_t e = a * 3 + 1, f = b * 5 + 1, g = c * 7 + 1, h = d * 9 + 1;
    double x = a, y = 1.0000001, z = 0.5, w = b;
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }
        if (OP == 1) { a = a * b; c = c * d; e = e * f; g = g * h; b = b * a; d = d * c; f = f * e; h = h * g; }
        if (OP == 2) { a = __umulhi(a, b) | 1; c = __umulhi(c, d) | 1; e = __umulhi(e, f) | 1; g = __umulhi(g, h) | 1;
                       b = __umulhi(b, a) | 3; d = __umulhi(d, c) | 3; f = __umulhi(f, e) | 3; h = __umulhi(h, g) | 3; }
        if (OP == 3) { uint64_t p = (uint64_t)a * b + c; a = (uint32_t)p; c = (uint32_t)(p >> 32) | 1;
                       uint64_t r = (uint64_t)e * f + g; e = (uint32_t)r; g = (uint32_t)(r >> 32) | 1;
                       uint64_t s = (uint64_t)b * d + a; b = (uint32_t)s | 1; d = (uint32_t)(s >> 32) | 1;
                       uint64_t t = (uint64_t)f * h + e; f = (uint32_t)t | 1; h = (uint32_t)(t >> 32) | 1; }
        if (OP == 4) { x = fma(x, y, z); w = fma(w, y, z); z = fma(z, y, x); y = fma(y, 0.999, 1e-9); 
                       x = fma(x, y, w); w = fma(w, y, x); z = fma(z, 0.5, x); y = fma(y, 1.0, 1e-12);}
        if (OP == 6) { x = (double)(int)a + x; a = (uint32_t)(int)w; w = (double)(int)c + w; c = (uint32_t)(int)x;
                       z = (double)(int)e + z; e = (uint32_t)(int)y; y = (double)(int)g + z; g = (uint32_t)(int)z; }   // 4 cvt_f64_i32 + 4 cvt_i32_f64 + 4 add_f64
        if (OP == 7) { x = __builtin_rint(x * y) + z; w = __builtin_rint(w * y) + x; z = __builtin_rint(z * 0.37) + w; y = __builtin_rint(y * 1.7) + 0.3; }  // 4 mul + 4 rndne + 4 add
        if (OP == 5) { a = __mul24(a, b) + c; c = __mul24(c, d) + a; e = __mul24(e, f) + g; g = __mul24(g, h) + e;
                       b = __mul24(b, a) + d; d = __mul24(d, c) + b; f = __mul24(f, e) + h; h = __mul24(h, g) + f; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t)(x + w + z + y);
}
__global__ __launch_bounds__(256) void kmfma(int *out, int seed) {
    v4i a = {seed, seed + 1, (int)threadIdx.x, 3}, b = {5, (int)threadIdx.x, 7, seed};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < N_ITER; i++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 2048 * 4 * 4);
    const int blocks = 256 * 8;  // 8 waves per SIMD
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_fma_f64", "v_mad_u32_u24", "cvt f64<->i32 + add_f64 (12 ops)", "mul/rndne/add f64 (12 ops)"};
    double ops = (double)blocks * 256 * N_ITER * 8;
    float ms[8];
    ms[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[4] = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[5] = timeit([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[6] = timeit([&] { hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    ms[7] = timeit([&] { hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, 1u); });
    for (int i = 0; i < 8; i++) {
        double per = (i == 3) ? ops / 2 : (i >= 6 ? ops * 1.5 : ops);  // OP 3 issues 4 mads per iteration (others 8 ops)
        printf("%-16s %8.3f ms  %.2f Tlane-op/s\n", names[i], ms[i], per / ms[i] * 1e-9);
    }
    float m = timeit([&] { hipLaunchKernelGGL(kmfma, dim3(256 * 2), dim3(256), 0, 0, (int *)out, 1); });
    double mops = (double)256 * 2 * 4 * N_ITER * 4 * 32.0 * 32 * 32 * 2;
    printf("mfma_i32_32x32x32_i8 %8.3f ms  %.1f TOP/s\n", m, mops / m * 1e-9);
    return 0;
}
