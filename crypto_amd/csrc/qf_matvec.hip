// Kernel (c): HBM-streaming single-message products and the batched dot-product decryption.
// Byte-streaming kernels: every matrix / ciphertext byte is read exactly once with 16-byte-per-lane
// coalesced loads.  y = A*x streams the key's int8 digit planes (v_dot4c_i32_i8 against the digits
// of x); x^t*A and the decryption work on uint32 words with split 64-bit sums (low and high halves
// of each 62-bit product summed separately, so nothing can overflow) reduced once per output.
#include <cstdlib>
#include <mutex>
#include "qf_internal.h"

#define MV_BLOCK 256

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
// ---------------------------------------------------------------------------------------------
// y = A*x on the key's int8 DIGIT PLANES (the same [rb][ks][la][64 lanes][16 B] fragment image
// kernel (a) reads): ceil(bits(q)/8) bytes per matrix entry cross the HBM bus instead of the 4 of
// the uint32 copy (SIS n=512, q=8,380,417: 18.1 MB instead of 24.1 MB), every wave instruction is
// one contiguous 1-KiB fragment, and a workgroup owns a short run of k-steps of ONE row block, so
// the SIS matrix becomes ~500 workgroups with every byte requested within the first microsecond.
//
//   workgroup (4 waves) = row block rbi x k-steps [ks0, ks0 + 4U): wave w owns k-steps ks0+4u+w.
//   1. every lane issues its U*L fragment loads (16 B each) at once;
//   2. meanwhile thread t recodes x[32 ks0 + t] into L balanced digits -> LDS digit rows;
//   3. per fragment: L*L*4 v_dot4c_i32_i8 against the x digits of its 16 columns (digit pairs of
//      equal weight share an int32 sum, as in the GEMM); recombined with the centred 2^(8s) mod q,
//      reduced, summed over the two lane halves and the four waves -> one residue per row;
//   4. cross-workgroup sum WITHOUT a second kernel and without a ticket: one returning 64-bit
//      atomic add per row of (1 << 48 | residue) into the scratch word of that row; the add that
//      finds nsplit-1 earlier arrivals in the top 16 bits holds the complete sum, writes y[row]
//      and swaps the word back to 0 -- the scratch is all-zero again when the kernel ends.  Data
//      moves between workgroups through device-scope atomics only, so nothing depends on
//      placement, dispatch order or cache state (MI355X_MICROARCH.md, inter-workgroup visibility).
// Scratch: one uint64 per padded row, zero before the launch (qf_ctx owns one, zeroed when it is
// allocated; the no-allocation entry point uses the file-scope array below).
// ---------------------------------------------------------------------------------------------
#define MVP_THREADS 256
struct QfMvpArgs {
    const int8_t *planes;
    const uint32_t *x;
    uint32_t *y;
    unsigned long long *acc;
    uint32_t rb, ks, rows, cols, nsplit;
    int32_t cs[8];
    QfMod mod;
};

template <int L, int U>
__global__ __launch_bounds__(MVP_THREADS) void qf_matvec_planes_kernel(QfMvpArgs a) {
    constexpr int KW = 4 * U;                       // k-steps per workgroup
    constexpr int S = 2 * L - 1;
    __shared__ __attribute__((aligned(16))) uint8_t xd[L][KW * 32];
    __shared__ uint32_t part[4][32];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, h = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rbi = blockIdx.x / a.nsplit, sp = blockIdx.x % a.nsplit;
    const uint32_t ks0 = sp * KW;

    // x is requested FIRST: vector loads return in order, so its recode then runs while the fragments
    // are still on their way instead of behind them
    constexpr int XPT = (KW * 32 + MVP_THREADS - 1) / MVP_THREADS;   // x values per thread (1 or 2)
    uint32_t xraw[XPT];
#pragma unroll
    for (int j = 0; j < XPT; j++) {
        const uint32_t t = tid + j * MVP_THREADS, k = ks0 * 32 + t;
        xraw[j] = (t < KW * 32 && k < a.cols) ? a.x[k] : 0u;
    }
    // 1. the fragments (k-steps past the key's padded width are skipped: wave-uniform)
    v4i kf[U][L];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t ksi = ks0 + 4 * u + wave;
        const v4i *src = (const v4i *)(a.planes + (((size_t)rbi * a.ks + (ksi < a.ks ? ksi : a.ks - 1)) * L * 64 + lane) * 16);
#pragma unroll
        for (int l = 0; l < L; l++) kf[u][l] = src[64 * l];
    }
    // 2. this workgroup's slice of x as balanced digits (same recode as qf_recode_kernel)
#pragma unroll
    for (int j = 0; j < XPT; j++) {
        const uint32_t t = tid + j * MVP_THREADS;
        if (t >= KW * 32) break;
        // entries past the key's columns arrive as 0 (they only meet zero-padded key digits anyway)
        const uint32_t c = qf_red_u64(xraw[j], a.mod);
        int64_t maxpos = 0;
#pragma unroll
        for (int i = 0; i < L; i++) maxpos = maxpos * 256 + 127;
        int64_t rep = (int64_t)c <= maxpos ? (int64_t)c : (int64_t)c - (int64_t)a.mod.q;
#pragma unroll
        for (int l = 0; l < L; l++) {
            const int64_t d = ((rep + 128) & 255) - 128;
            xd[l][t] = (uint8_t)(d & 255);
            rep = (rep - d) >> 8;
        }
    }
    __syncthreads();
    // 3. digit products
    int acc[S];
#pragma unroll
    for (int s = 0; s < S; s++) acc[s] = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t ksi = ks0 + 4 * u + wave;
        if (ksi >= a.ks) continue;
#pragma unroll
        for (int lb = 0; lb < L; lb++) {
            const v4i xs = *(const v4i *)&xd[lb][(4 * u + wave) * 32 + 16 * h];
#pragma unroll
            for (int la = 0; la < L; la++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[la + lb] = __builtin_amdgcn_sdot4(kf[u][la][i], xs[i], acc[la + lb], false);
        }
    }
    int64_t v = 0;
#pragma unroll
    for (int s = 0; s < S; s++) v += (int64_t)acc[s] * (int64_t)a.cs[s];   // |acc| < 2^25, |cs| < 2^30
    uint32_t r = qf_red_i64(v, a.mod);
    r = qf_addmod(r, (uint32_t)__shfl_xor((int)r, 32, 64), a.mod.q);
    if (lane < 32) part[wave][lane] = r;
    __syncthreads();
    if (tid < 32) {
        r = qf_addmod(qf_addmod(part[0][tid], part[1][tid], a.mod.q), qf_addmod(part[2][tid], part[3][tid], a.mod.q), a.mod.q);
        const uint32_t row = rbi * 32 + tid;
        // 4. (padded rows take part too: they are what returns their scratch words to zero)
        uint64_t total = r;
        bool last = true;
        if (a.nsplit > 1) {
            unsigned long long *w = a.acc + row;
            const unsigned long long old = __hip_atomic_fetch_add(w, (1ull << 48) | (unsigned long long)r,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (uint32_t)(old >> 48) == a.nsplit - 1;
            total = (old & 0xFFFFFFFFFFFFull) + r;
            if (last) (void)__hip_atomic_exchange(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last && row < a.rows) a.y[row] = qf_red_u64(total, a.mod);
    }
}

// ---- the scratch of the no-allocation entry point (include/qfgpu.h: qf_matvec_modq_static_dev) ----
// One uint64 per padded key row.  In HOST code the array's name is only a stand-in for the symbol:
// its device address is looked up once per device (qf_static_scratch_init, called by qf_ctx_create
// with that device current) and only that pointer is ever handed to a kernel or a memset.
#define QF_STATIC_ROWS 32768
static __device__ __align__(256) unsigned long long qf_static_scratch[QF_STATIC_ROWS];
static unsigned long long *g_static_ptr[QF_MAX_DEVICES];
static std::mutex g_static_mu;   // two threads may create the first contexts of one device at once

int qf_static_scratch_init(qf_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= QF_MAX_DEVICES) return qf_fail(ctx, QF_ERR_NOTSUP, "static scratch: device index out of range");
    std::lock_guard<std::mutex> lk(g_static_mu);
    if (g_static_ptr[ctx->device]) return QF_OK;
    unsigned long long *p = nullptr;
    QF_HIP(ctx, hipGetSymbolAddress((void **)&p, HIP_SYMBOL(qf_static_scratch)));
    if (!p) return qf_fail(ctx, QF_ERR_DEVICE, "static scratch: hipGetSymbolAddress returned null");
    g_static_ptr[ctx->device] = p;
    return QF_OK;
}

template <int L>
static int mvp_launch(qf_ctx *ctx, const QfMvpArgs &a0, uint32_t ks) {
    // k-steps per wave: the smallest U that brings the grid to <= 768 workgroups.  Measured on the
    // cold SIS key (profiles/r2/matvec_variants.log): 1,472 / 736 / 496 / 368 workgroups take 7.1 / 6.1 /
    // 6.1 / 6.4 us -- fewer arrivals per scratch word (the atomics of one row block contend on two
    // 128-byte lines) against fewer, longer workgroups.  Default cache policy: non-temporal loads
    // were 0.1-0.3 us slower here and would keep a re-used key out of the Infinity Cache.
    int U = 1;
    while (U < 4 && (uint64_t)a0.rb * ((ks + 4 * U - 1) / (4 * U)) > 768) U++;
    if (const char *e = getenv("QF_MATVEC_U")) { int v = atoi(e); if (v >= 1 && v <= 4) U = v; }   // experiments
    QfMvpArgs a = a0;
    a.nsplit = (ks + 4 * U - 1) / (4 * U);
    if (a.nsplit >= 65536) return qf_fail(ctx, QF_ERR_NOTSUP, "matvec: more than 2^16 k-chunks per row block");
    const dim3 grid(a.rb * a.nsplit), block(MVP_THREADS);
    QfTimer t(ctx, &ctx->ms_matvec);
    switch (U) {
        case 1: hipLaunchKernelGGL((qf_matvec_planes_kernel<L, 1>), grid, block, 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL((qf_matvec_planes_kernel<L, 2>), grid, block, 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL((qf_matvec_planes_kernel<L, 3>), grid, block, 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL((qf_matvec_planes_kernel<L, 4>), grid, block, 0, ctx->stream, a); break;
    }
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// acc: rb*32 zeroed uint64 words on the key's device
static int mvp_run(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y, unsigned long long *acc) {
    QfMvpArgs a;
    a.planes = key->planes; a.x = x; a.y = y; a.acc = acc;
    a.rb = key->rb; a.ks = key->ks; a.rows = key->rows; a.cols = key->cols; a.nsplit = 1;
    a.mod = key->mod;
    uint64_t p = 1 % key->q;
    for (int s = 0; s < 8; s++) {
        int64_t c = (int64_t)p;
        if (c > (int64_t)(key->q / 2)) c -= key->q;
        a.cs[s] = (int32_t)c;
        p = (p * 256) % key->q;
    }
    if ((uint64_t)key->rb * ((key->ks + 3) / 4) > 0x7fffffffull) return qf_fail(ctx, QF_ERR_NOTSUP, "matvec: key too large");
    switch (key->la) {
        case 1: return mvp_launch<1>(ctx, a, key->ks);
        case 2: return mvp_launch<2>(ctx, a, key->ks);
        case 3: return mvp_launch<3>(ctx, a, key->ks);
        case 4: return mvp_launch<4>(ctx, a, key->ks);
    }
    return qf_fail(ctx, QF_ERR_NOTSUP, "matvec: digit count not instantiated");
}

int qf_launch_matvec(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y) {
    const size_t need = (size_t)key->rb * 32 * sizeof(unsigned long long);
    if (need > ctx->mv_acc_bytes) {   // grown on demand, zeroed once: the kernel leaves it zero
        if (ctx->mv_acc) { QF_HIP(ctx, hipStreamSynchronize(ctx->stream)); QF_HIP(ctx, hipFree(ctx->mv_acc)); ctx->mv_acc = nullptr; ctx->mv_acc_bytes = 0; }
        QF_HIP(ctx, hipMalloc(&ctx->mv_acc, need));
        ctx->mv_acc_bytes = need;
        QF_HIP(ctx, hipMemsetAsync(ctx->mv_acc, 0, need, ctx->stream));
        ctx->mv_acc_dirty = false;
    }
    // The kernel returns every word it used to zero -- IF it ran to its end.  After a launch that failed
    // (or was refused) the words may hold half a handshake: clear them before they are trusted again.
    if (ctx->mv_acc_dirty) {
        QF_HIP(ctx, hipMemsetAsync(ctx->mv_acc, 0, ctx->mv_acc_bytes, ctx->stream));
        ctx->mv_acc_dirty = false;
    }
    const int rc = mvp_run(ctx, key, x, y, (unsigned long long *)ctx->mv_acc);
    if (rc) ctx->mv_acc_dirty = true;
    return rc;
}

// No allocation anywhere on this path: the cross-workgroup sums live in qf_static_scratch (through
// the per-device pointer).  The words it uses are cleared in front of every launch, so a call never
// depends on what an earlier (possibly aborted) call left behind.
int qf_launch_matvec_static(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y) {
    if (ctx->device < 0 || ctx->device >= QF_MAX_DEVICES || !g_static_ptr[ctx->device])
        return qf_fail(ctx, QF_ERR_DEVICE, "matvec (static): scratch address of this device was not resolved");
    if ((uint64_t)key->rb * 32 > QF_STATIC_ROWS)
        return qf_fail(ctx, QF_ERR_NOTSUP, "matvec (static): key has more than 32768 (padded) rows");
    unsigned long long *acc = g_static_ptr[ctx->device];
    QF_HIP(ctx, hipMemsetAsync(acc, 0, (size_t)key->rb * 32 * sizeof(unsigned long long), ctx->stream));
    return mvp_run(ctx, key, x, y, acc);
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
                                 QfMod mod, const uint32_t *__restrict__ addend, uint32_t *__restrict__ y) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    uint64_t lo = 0, hi = 0;
    for (uint32_t c = 0; c < nchunk; c++) { lo += partial[(size_t)c * cols + j].lo; hi += partial[(size_t)c * cols + j].hi; }
    uint32_t v = acc_finish(lo, hi, mod);
    if (addend) v = qf_red_u64((uint64_t)v + addend[j], mod);
    y[j] = v;
}

// y^t = x^t*A (+ addend^t) for a plain row-major uint32 matrix on the device (no key object)
int qf_launch_vecmat_raw(qf_ctx *ctx, const uint32_t *A, uint32_t rows, uint32_t cols, const QfMod &mod,
                         const uint32_t *x, const uint32_t *addend, uint32_t *y) {
    uint32_t nchunk = (rows + VM_ROWS - 1) / VM_ROWS;
    int rc = qf_ws2_reserve(ctx, (size_t)cols * nchunk * sizeof(Acc2));
    if (rc) return rc;
    dim3 grid((cols + MV_BLOCK * 4 - 1) / (MV_BLOCK * 4), nchunk), block(MV_BLOCK);
    {
        QfTimer t(ctx, &ctx->ms_matvec);
        hipLaunchKernelGGL(qf_vecmat_partial, grid, block, 0, ctx->stream, A, rows, cols, x, (Acc2 *)ctx->ws2);
    }
    hipLaunchKernelGGL(qf_vecmat_finish, dim3((cols + 255) / 256), dim3(256), 0, ctx->stream,
                       (const Acc2 *)ctx->ws2, cols, nchunk, mod, addend, y);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
int qf_launch_vecmat(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y) {
    return qf_launch_vecmat_raw(ctx, key->u32, key->rows, key->cols, key->mod, x, nullptr, y);
}

// Batched decryption: a wave handles DEC_U = 8 ciphertexts per iteration (all their loads in flight
// together), lanes stride over the n entries.  The 8 per-lane sums are then reduced across the
// wave TOGETHER by a transposing butterfly: three halving steps (lane^32, ^16, ^8: a lane keeps half
// of its sums and receives the partner's copies of those), after which each lane owns ONE ciphertext
// (u = lane bits 5:3), and three plain steps (^4, ^2, ^1) -- 7 + 3 exchanged values per lane instead
// of 8 x 6, and ONE reduction mod q + decision per 8 ciphertexts instead of 8.  SINGLE: when
// (n+1)(q-1)^2 < 2^64 the products go into one 64-bit sum per ciphertext (one v_mad_u64_u32 per
// entry); otherwise low and high halves of the 62-bit products are summed separately.
#define DEC_U 8
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}
// v[0..7] per lane -> the wave total of v[u], u = lane bits 5:3, in every lane of that group of 8
__device__ __forceinline__ uint64_t wave_transpose_sum8(uint64_t (&v)[DEC_U], int lane) {
    {
        const bool up = lane & 32;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t keep = up ? v[k + 4] : v[k], give = up ? v[k] : v[k + 4];
            v[k] = keep + shfl_xor_u64(give, 32);
        }
    }
    {
        const bool up = lane & 16;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint64_t keep = up ? v[k + 2] : v[k], give = up ? v[k] : v[k + 2];
            v[k] = keep + shfl_xor_u64(give, 16);
        }
    }
    const bool up = lane & 8;
    uint64_t t = (up ? v[1] : v[0]) + shfl_xor_u64(up ? v[0] : v[1], 8);
    t += shfl_xor_u64(t, 4);
    t += shfl_xor_u64(t, 2);
    t += shfl_xor_u64(t, 1);
    return t;
}
// 16-byte loads that need only 4-byte alignment (ciphertext rows are 4(n+1) bytes apart: 1,028 at n = 256)
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// One group of DEC_U ciphertexts (rows c[u]) for one wave; returns the group's residues (lane group u
// = lane bits 5:3 holds r_u).  SINGLE also reports, wave-uniformly, whether any word of w or of the
// ciphertexts had a bit at or above `cbits` set -- then its 64-bit sums may have wrapped and the caller redoes the
// group with split sums.
template <bool SINGLE>
__device__ __forceinline__ uint32_t dec_group(const uint32_t *__restrict__ w, uint32_t n, const QfMod &mod,
                                              const uint32_t *const (&c)[DEC_U], int lane, uint32_t cbits, bool &wide) {
    const uint32_t nvec = n & ~3u;       // columns covered by whole 4-word groups
    uint64_t lo[DEC_U], hi[SINGLE ? 1 : DEC_U];
    uint32_t seen = 0;
#pragma unroll
    for (int u = 0; u < DEC_U; u++) {
        lo[u] = 0;
        if (!SINGLE) hi[u] = 0;
    }
#define DEC_MAD(u_, wv_, cv_) {                                                                   \
    const uint64_t p_ = (uint64_t)(wv_) * (cv_);                                                  \
    if (SINGLE) lo[u_] += p_; else { lo[u_] += (uint32_t)p_; hi[u_] += p_ >> 32; } }
    // a lane takes 4 adjacent columns per step (one 16-byte load per ciphertext: a whole 1-KiB run per
    // wave instruction, DEC_U of them in flight), the same 4 entries of w serve all DEC_U ciphertexts
    for (uint32_t i = 4 * lane; i < nvec; i += 256) {
        const u32x4_a4 wv = *(const u32x4_a4 *)(w + i);
        u32x4_a4 cv[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; u++) cv[u] = *(const u32x4_a4 *)(c[u] + i);
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            DEC_MAD(u, wv.x, cv[u].x) DEC_MAD(u, wv.y, cv[u].y)
            DEC_MAD(u, wv.z, cv[u].z) DEC_MAD(u, wv.w, cv[u].w)
            if (SINGLE) seen |= cv[u].x | cv[u].y | cv[u].z | cv[u].w;
        }
        if (SINGLE) seen |= wv.x | wv.y | wv.z | wv.w;
    }
    // the n mod 4 trailing columns and the ciphertext's last entry c[n] (coefficient 1): lanes 0..3
    if ((uint32_t)lane <= n - nvec) {
        const uint32_t i = nvec + lane;              // i == n: the last entry
        const uint32_t wt = i < n ? w[i] : 1u;
        if (SINGLE) seen |= wt;
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            const uint32_t cvs = c[u][i];
            DEC_MAD(u, wt, cvs)
            if (SINGLE) seen |= cvs;
        }
    }
#undef DEC_MAD
    const uint64_t tl = wave_transpose_sum8(lo, lane);
    if constexpr (SINGLE) {
        wide = cbits < 32 && __any((int)((seen >> cbits) != 0));
        return qf_red_u64(tl, mod);
    } else {
        wide = false;
        const uint64_t th = wave_transpose_sum8(hi, lane);
        return acc_finish(tl, th, mod);
    }
}
// cbits > 0: one 64-bit sum per ciphertext (exact while every word of w and C is below 2^cbits: the
// launcher has checked (n+1)(2^cbits-1)^2 < 2^64 with 2^cbits >= q, so canonical operands always
// qualify); cbits == 0: split sums throughout.
__global__ __launch_bounds__(256) void qf_dot_decode_kernel(const uint32_t *__restrict__ w, uint32_t n,
                                                            QfMod mod, const uint32_t *__restrict__ C,
                                                            uint64_t B, uint8_t *__restrict__ bits, uint32_t cbits) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t stride = n + 1;
    for (uint64_t b0 = wave * DEC_U; b0 < B; b0 += nwaves * DEC_U) {
        const uint32_t *c[DEC_U];
#pragma unroll
        for (int u = 0; u < DEC_U; u++) {
            const uint64_t b = b0 + u < B ? b0 + u : B - 1;  // clamped: the duplicate result is dropped
            c[u] = C + b * stride;
        }
        bool wide = true;
        uint32_t r = 0;
        if (cbits) r = dec_group<true>(w, n, mod, c, lane, cbits, wide);
        if (wide) r = dec_group<false>(w, n, mod, c, lane, cbits, wide);
        const uint32_t u = (uint32_t)lane >> 3;
        if ((lane & 7) == 0 && b0 + u < B) bits[b0 + u] = (uint8_t)qf_decide_bit(r, mod);
    }
}
int qf_launch_dot_decode(qf_ctx *ctx, const uint32_t *w, uint32_t n, const QfMod &mod, const uint32_t *C,
                         uint64_t B, uint8_t *bits) {
    uint64_t blocks = (B + 4 * DEC_U - 1) / (4 * DEC_U);  // 4 waves per block, DEC_U ciphertexts per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks == 0) return QF_OK;
    // one 64-bit sum per ciphertext is exact while (n+1) products of (2^cbits - 1)^2 stay below 2^64,
    // cbits = bit length of q (canonical words are below 2^cbits; the kernel checks the words it reads
    // and redoes a group with split sums if one is not)
    uint32_t cbits = 0;
    while ((1ull << cbits) < mod.q) cbits++;
    const long double worst = ((long double)n + 1) * ((long double)(1ull << cbits) - 1) * ((long double)(1ull << cbits) - 1);
    if (!(worst < 18446744073709551615.0L)) cbits = 0;
    if (const char *e = getenv("QF_DECODE_SPLIT")) if (atoi(e)) cbits = 0;   // tests: force the split sums
    QfTimer t(ctx, &ctx->ms_decode);
    hipLaunchKernelGGL(qf_dot_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w, n, mod, C, B, bits, cbits);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}
