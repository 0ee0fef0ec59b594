// Host-side internals of libqfgpu: handle structs, error plumbing, launch prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include "../../include/qfgpu.h"
#include "qf_device.h"

struct qf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scratch (grown on demand, owned by the context)
    void *ws = nullptr;       // recoded batch operand planes
    size_t ws_bytes = 0;
    void *ws2 = nullptr;      // vec-mat partial sums
    size_t ws2_bytes = 0;
    void *mv_acc = nullptr;   // mat-vec cross-workgroup sums: one uint64 per padded row, all-zero between launches
    size_t mv_acc_bytes = 0;
    bool mv_acc_dirty = false; // a launch over mv_acc failed: its words are cleared before the next one
    uint32_t *hostvec = nullptr;  // staging of the *_host single-message entry points (x then y)
    size_t hostvec_words = 0;
    uint32_t *flag = nullptr; // device error flag (range violations)
    uint32_t *hflag = nullptr; // pinned host copy of it + the event behind the copy (digit-promise check of qf_matmul_modq_dev)
    hipEvent_t evf = nullptr;
    // discrete-Gaussian inversion table of the last (n, s) asked for (qf_sample_discrete_gauss_dev)
    uint64_t *cdt = nullptr;
    uint32_t cdt_len = 0, cdt_n = 0;
    double cdt_s = 0.0;
    // timing
    int timing = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float ms_gemm = -1.f, ms_ntt = -1.f, ms_matvec = -1.f, ms_decode = -1.f;
};

// per-device one-time set-up (function attributes such as the dynamic LDS limit are per device: a
// process that drives several GPUs must set them on each).  Several contexts / threads may meet
// here for one device: the work runs under a lock, and a device is marked only after its call
// returned hipSuccess -- a failed attempt is retried by the next launch, and no thread can launch
// with > 64 KiB of dynamic LDS before the attribute is in place.
#define QF_MAX_DEVICES 64
struct QfPerDevice {
    std::atomic<bool> done[QF_MAX_DEVICES] = {};
    std::mutex mu;
    template <class F>
    hipError_t once(int device, F set) {
        if (device < 0 || device >= QF_MAX_DEVICES) return set();   // beyond the table: every time
        if (done[device].load(std::memory_order_acquire)) return hipSuccess;
        std::lock_guard<std::mutex> lk(mu);
        if (done[device].load(std::memory_order_relaxed)) return hipSuccess;
        const hipError_t e = set();
        if (e == hipSuccess) done[device].store(true, std::memory_order_release);
        return e;
    }
};

struct qf_key {
    int device = 0;
    uint32_t rows = 0, cols = 0, q = 0;
    int la = 0;         // signed 8-bit digits per entry
    uint32_t rb = 0;    // row blocks of 32
    uint32_t ks = 0;    // k-steps of 32 (padded to a multiple of 4)
    void *block = nullptr;     // the key's one device allocation: u32 | planes | rem_planes
    size_t block_bytes = 0;
    int8_t *planes = nullptr;  // [rb][ks][la][64 lanes][16 B]   (MFMA fragment order)
    uint32_t *u32 = nullptr;   // rows x cols row-major copy
    // packed remainder block (bit-operand products): when r = rows mod 32 satisfies 0 < r*la <= 32
    // the r trailing rows' digits fill ONE fragment per k-step, [ks][64 lanes][16 B]
    int8_t *rem_planes = nullptr;
    uint32_t rem_rows = 0;
    // folded planes for an LB-digit batch operand (built on first use): k-step ks*LB + b holds the
    // digits of 256^b * A mod q, so A*x = [A | 256A | ...] * [x_0 ; x_1 ; ...] runs as ONE
    // single-digit product over LB*K columns.  [rb][ks*LB][la][64 lanes][16 B]
    mutable int8_t *fold_planes[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    QfMod mod;
};

struct qf_ring {
    int device = 0;
    uint32_t n = 0, q = 0, lg = 0;
    int ntt = 0;           // 1 if q prime and 2n | q-1
    int wave = 0;          // 1 if the wave-per-polynomial kernels (qf_ntt_wave.hip) serve this ring
    QfMod mod;
    // twiddles: psi^bitrev(i) (forward), psi^-bitrev(i) (inverse) and their Shoup companions
    uint32_t *tw = nullptr;   // [4][n]: fw, fw', iw, iw'
    uint32_t ninv = 0, ninv_p = 0;
};

struct qf_rkey {
    int device = 0;
    uint32_t n = 0, q = 0, npoly = 0;
    int ntt = 0, wave = 0;
    uint32_t *coef = nullptr;  // npoly x n canonical coefficients
    uint32_t *hat = nullptr;   // npoly x n NTT-domain values (ntt rings only); wave rings: npoly x n
                               // interleaved (value, Shoup companion) pairs in register order
    uint32_t *hat_p = nullptr; // Shoup companions of hat (null on wave rings)
};

#define QF_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            if (ctx) (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);   \
            return QF_ERR_DEVICE;                                                      \
        }                                                                              \
    } while (0)

static inline int qf_fail(qf_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

QfMod qf_make_mod(uint32_t q);
int qf_ws_reserve(qf_ctx *ctx, size_t bytes);
int qf_ws2_reserve(qf_ctx *ctx, size_t bytes);

// ---- launchers implemented in the .hip files ----
// qf_gemm.hip
int qf_digits_for_modulus(uint32_t q);
int qf_launch_recode(qf_ctx *ctx, const void *src, int kind, size_t stride_bytes, uint32_t rows,
                     uint32_t cols, const QfMod &mod, int digits, uint32_t ks_pad, int8_t *planes,
                     uint32_t *flag, uint32_t mult = 1, uint32_t fold = 1, uint32_t fold_b = 0);
int qf_launch_recode_rem(qf_ctx *ctx, const uint32_t *src, uint32_t cols, uint32_t first_row, uint32_t rem_rows,
                         const QfMod &mod, int digits, uint32_t ks_pad, int8_t *planes);
int qf_launch_gemm(qf_ctx *ctx, const qf_key *key, const void *X, int x_is_bits, size_t x_stride,
                   int lb, uint64_t B, const uint32_t *addend, const uint8_t *msg, uint32_t *out);
int qf_launch_transpose_u32(qf_ctx *ctx, const uint32_t *src, uint32_t rows, uint32_t cols, uint32_t *dst);
// qf_matvec.hip
int qf_launch_matvec(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y);
int qf_launch_vecmat(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y);
int qf_launch_vecmat_raw(qf_ctx *ctx, const uint32_t *A, uint32_t rows, uint32_t cols, const QfMod &mod,
                         const uint32_t *x, const uint32_t *addend, uint32_t *y);
int qf_launch_matvec_static(qf_ctx *ctx, const qf_key *key, const uint32_t *x, uint32_t *y);
int qf_static_scratch_init(qf_ctx *ctx);   // resolves the per-device address of the static scratch array
int qf_launch_dot_decode(qf_ctx *ctx, const uint32_t *w, uint32_t n, const QfMod &mod, const uint32_t *C,
                         uint64_t B, uint8_t *bits);
// qf_sampler.hip
int qf_launch_sample_uniform(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, uint32_t q, uint32_t *out);
int qf_launch_sample_bits_kt(qf_ctx *ctx, QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols, size_t kt_stride, uint8_t *out);
int qf_launch_sample_bits(qf_ctx *ctx, QfGen base, uint64_t first_row, uint64_t rows, uint32_t cols,
                          uint32_t stride_bytes, uint8_t *out);
int qf_launch_sample_cdt(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, const QfMod &mod,
                         const uint64_t *thr_dev, uint32_t len, int64_t lo, uint32_t *out);
int qf_launch_sample_noise(qf_ctx *ctx, QfGen base, uint64_t first, uint64_t count, const QfMod &mod,
                           uint32_t W, uint32_t *out);
// qf_ntt.hip
int qf_launch_ntt_forward_keys(qf_ctx *ctx, const qf_ring *ring, qf_rkey *rk);
int qf_launch_polymul(qf_ctx *ctx, const qf_ring *ring, const uint32_t *a, const uint32_t *b, uint64_t B, uint32_t *c);
int qf_launch_ringlpr_enc(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *pk, const uint32_t *r,
                          const uint32_t *e1, const uint32_t *e2, const uint8_t *mu, uint64_t B,
                          uint32_t *u, uint32_t *v);
int qf_launch_ringlpr_dec(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *sk, const uint32_t *u,
                          const uint32_t *v, uint64_t B, uint8_t *mu);
int qf_launch_fdh_verify(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *a, const int32_t *sigma,
                         const uint32_t *u, uint64_t norm_bound, uint64_t B, uint32_t *t, uint8_t *ok);

// qf_sha256.hip
int qf_launch_hash_to_zq(qf_ctx *ctx, const uint8_t *msgs, const uint64_t *offs, uint64_t B, uint32_t rows,
                         uint32_t cols, const QfMod &mod, uint32_t *out);

// timing helpers: bracket the dominant kernel of an op with events when ctx->timing is set
struct QfTimer {
    qf_ctx *ctx;
    float *slot;
    QfTimer(qf_ctx *c, float *s) : ctx(c), slot(s) {
        if (ctx->timing) hipEventRecord(ctx->ev0, ctx->stream);
    }
    ~QfTimer() {
        if (ctx->timing) {
            hipEventRecord(ctx->ev1, ctx->stream);
            hipEventSynchronize(ctx->ev1);
            hipEventElapsedTime(slot, ctx->ev0, ctx->ev1);
        }
    }
};
