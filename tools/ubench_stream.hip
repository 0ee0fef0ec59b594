// Floor for kernel (c): how fast can the chip read 18.1 MB that is cold in every cache, in ONE launch?
// Same grid shapes as qf_matvec_planes_kernel (256 threads, U*3 16-byte loads per lane, all issued
// at once), no arithmetic beyond an OR-reduction, no cross-workgroup step.  24 buffers are rotated
// (434 MB > the 256 MiB Infinity Cache).  Prints kernel time from HIP events around single launches
// and the back-to-back rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NL>
__global__ __launch_bounds__(256) void rd(const v4i *__restrict__ p, int *out) {
    const size_t base = (size_t)blockIdx.x * 256 * NL;   // in v4i units: NL KiB*... contiguous per WG
    v4i v[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) v[i] = p[base + (size_t)i * 256 + threadIdx.x];
    int o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= v[i][0] | v[i][1] | v[i][2] | v[i][3];
    if (o == 0x12345678) out[blockIdx.x] = o;
}
int main() {
    const size_t bytes = 512ull * 11776 * 3;   // 18,087,936
    const int NB = 24;
    std::vector<void *> buf(NB);
    for (auto &b : buf) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
    int *out; hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int nl, void *b) {
        const unsigned grid = (unsigned)(bytes / (256 * 16 * nl));
        switch (nl) {
            case 3: hipLaunchKernelGGL(rd<3>, dim3(grid), dim3(256), 0, 0, (const v4i *)b, out); break;
            case 6: hipLaunchKernelGGL(rd<6>, dim3(grid), dim3(256), 0, 0, (const v4i *)b, out); break;
            case 12: hipLaunchKernelGGL(rd<12>, dim3(grid), dim3(256), 0, 0, (const v4i *)b, out); break;
        }
    };
    for (int nl : {3, 6, 12}) {
        for (int i = 0; i < NB; i++) run(nl, buf[i]);
        hipDeviceSynchronize();
        std::vector<float> t;
        for (int r = 0; r < 4 * NB; r++) {
            hipEventRecord(e0); run(nl, buf[r % NB]); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        hipEventRecord(e0);
        for (int r = 0; r < 8 * NB; r++) run(nl, buf[r % NB]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("loads/lane %2d  grid %5zu  single-launch us: min %.2f median %.2f   back-to-back us/launch %.2f  (%.0f GB/s)\n", nl,
               bytes / (256 * 16 * nl), t[0], t[t.size() / 2], ms * 1e3 / (8 * NB), bytes / (ms * 1e3 / (8 * NB)) / 1e3);
    }
    return 0;
}
