// Diagnostic: sustained int8 MFMA rate on gfx950 with realistic operand data (random digits x {0,1}
// bits or random x random), both shapes, operands in registers, several ms per launch so that the
// clock the chip holds under load shows.  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int SHAPE, int DATA>
__global__ __launch_bounds__(256) void kmfma(int *out, int iters, unsigned long long *clk) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        uint32_t ra = mix(t * 32 + i * 4 + j), rb = mix(t * 32 + 16 + i * 4 + j);
        a[i][j] = DATA == 0 ? 0 : DATA == 1 ? (int)(ra & 0x01010101u) : (int)ra;   // zero / bits / random
        b[i][j] = DATA == 0 ? 0 : (int)rb;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    if (SHAPE == 32) {
        v16i c[4] = {{0}, {0}, {0}, {0}};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 4; k++) c[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], b[(u + k) & 3], c[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; k++) for (int r = 0; r < 16; r++) s += c[k][r];
    } else {
        v4i c[8];
        for (int k = 0; k < 8; k++) c[k] = (v4i){0, 0, 0, 0};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 8; k++) c[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], b[(u + k) & 3], c[k], 0, 0, 0);
        }
        for (int k = 0; k < 8; k++) for (int r = 0; r < 4; r++) s += c[k][r];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int SHAPE, int DATA>
void run(const char *name, int *out, unsigned long long *clk, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = SHAPE == 32 ? 6000 : 6000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((kmfma<SHAPE, DATA>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((kmfma<SHAPE, DATA>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double nm = SHAPE == 32 ? 16.0 : 32.0, opsper = SHAPE == 32 ? 32.0 * 32 * 32 * 2 : 16.0 * 16 * 64 * 2;
    double ops = (double)blocks * 4 * iters * nm * opsper;
    printf("%-28s waves/SIMD %d  %7.3f ms  %7.1f TOP/s  clock %.2f GHz  cyc/mfma %.1f\n", name, waves_per_simd, ms, ops / ms * 1e-9,
           (double)h[0] / (double)h[1] * 0.1, (double)h[0] / (iters * nm) / waves_per_simd);
}
int main() {
    int *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 16);
    run<32, 0>("32x32x32 zeros", out, clk, 1);
    run<32, 1>("32x32x32 bits x random", out, clk, 1);
    run<32, 2>("32x32x32 random x random", out, clk, 1);
    run<32, 1>("32x32x32 bits x random", out, clk, 2);
    run<16, 0>("16x16x64 zeros", out, clk, 1);
    run<16, 1>("16x16x64 bits x random", out, clk, 1);
    run<16, 2>("16x16x64 random x random", out, clk, 1);
    run<16, 1>("16x16x64 bits x random", out, clk, 2);
    return 0;
}
