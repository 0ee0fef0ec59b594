// libqfgpu C ABI (include/qfgpu.h): argument checking, handle management, host-side number
// theory (primality, roots of unity, twiddle tables).  All arithmetic on bulk data happens in
// the kernels of qf_gemm.hip / qf_matvec.hip / qf_ntt.hip / qf_sampler.hip.
#include <cstring>
#include <new>
#include <vector>
#include <mutex>
#include <vector>
#include "qf_internal.h"
#include "qf_ring.h"

// ------------------------------------------------------------------ small host number theory
static uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t m) { return (unsigned __int128)a * b % m; }
static uint64_t h_powmod(uint64_t a, uint64_t e, uint64_t m) {
    uint64_t r = 1 % m;
    a %= m;
    while (e) { if (e & 1) r = h_mulmod(r, a, m); a = h_mulmod(a, a, m); e >>= 1; }
    return r;
}
static bool h_is_prime(uint32_t n) {
    if (n < 2) return false;
    for (uint32_t p : {2u, 3u, 5u, 7u, 11u, 13u}) { if (n % p == 0) return n == p; }
    uint32_t d = n - 1; int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (uint32_t a : {2u, 7u, 61u}) {  // deterministic for n < 4,759,123,141
        uint64_t x = h_powmod(a % n, d, n);
        if (x == 0 || x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < r; i++) { x = h_mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}
static uint32_t h_bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

QfMod qf_make_mod(uint32_t q) {
    QfMod m;
    m.q = q; m.qhalf = q / 2;
    m.mu = (uint64_t)((((unsigned __int128)1) << 64) / q);
    uint64_t t = 1ull << 62;
    m.off = ((t + q - 1) / q) * q;
    m.r32 = (uint32_t)((1ull << 32) % q);
    m.pad = 0;
    return m;
}

int qf_ws_reserve(qf_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return QF_OK;
    if (ctx->ws) { QF_HIP(ctx, hipStreamSynchronize(ctx->stream)); QF_HIP(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    size_t want = bytes + bytes / 8;
    QF_HIP(ctx, hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    return QF_OK;
}
int qf_ws2_reserve(qf_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws2_bytes) return QF_OK;
    if (ctx->ws2) { QF_HIP(ctx, hipStreamSynchronize(ctx->stream)); QF_HIP(ctx, hipFree(ctx->ws2)); ctx->ws2 = nullptr; ctx->ws2_bytes = 0; }
    QF_HIP(ctx, hipMalloc(&ctx->ws2, bytes));
    ctx->ws2_bytes = bytes;
    return QF_OK;
}

static int check_q(qf_ctx *ctx, uint32_t q) {
    if (q < 2 || q >= (1u << 31)) return qf_fail(ctx, QF_ERR_MODULUS, "modulus must satisfy 2 <= q < 2^31");
    return QF_OK;
}
#define QF_BIND(ctx) QF_HIP(ctx, hipSetDevice((ctx)->device))

extern "C" {

const char *qf_version(void) { return "qfgpu 0.1 (gfx950)"; }

int qf_device_count(int *count) {
    if (!count) return QF_ERR_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { *count = 0; return QF_ERR_DEVICE; }
    *count = c;
    return QF_OK;
}

int qf_ctx_create(int device, void *stream, qf_ctx **out) {
    if (!out) return QF_ERR_ARG;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || device < 0 || device >= c) return QF_ERR_DEVICE;
    qf_ctx *ctx = new (std::nothrow) qf_ctx();
    if (!ctx) return QF_ERR_DEVICE;
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&ctx->flag, 256) != hipSuccess ||
        hipMemset(ctx->flag, 0, 256) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreateWithFlags(&ctx->evf, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&ctx->hflag, 64, hipHostMallocDefault) != hipSuccess) {
        if (ctx->flag) hipFree(ctx->flag);
        if (ctx->ev0) hipEventDestroy(ctx->ev0);
        if (ctx->ev1) hipEventDestroy(ctx->ev1);
        if (ctx->evf) hipEventDestroy(ctx->evf);
        delete ctx;
        return QF_ERR_DEVICE;
    }
    // per-device address of the no-allocation entry point's scratch (qf_matvec.hip), this device current
    if (qf_static_scratch_init(ctx) != QF_OK) { qf_ctx_destroy(ctx); return QF_ERR_DEVICE; }
    *out = ctx;
    return QF_OK;
}
int qf_ctx_set_stream(qf_ctx *ctx, void *stream) {
    if (!ctx) return QF_ERR_ARG;
    ctx->stream = (hipStream_t)stream;
    return QF_OK;
}
void qf_ctx_destroy(qf_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->ws2) hipFree(ctx->ws2);
    if (ctx->mv_acc) hipFree(ctx->mv_acc);
    if (ctx->hostvec) hipFree(ctx->hostvec);
    if (ctx->flag) hipFree(ctx->flag);
    if (ctx->cdt) hipFree(ctx->cdt);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->evf) hipEventDestroy(ctx->evf);
    if (ctx->hflag) hipHostFree(ctx->hflag);
    delete ctx;
}
int qf_sync(qf_ctx *ctx) {
    if (!ctx) return QF_ERR_ARG;
    QF_BIND(ctx);
    QF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QF_OK;
}
const char *qf_last_error(const qf_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int qf_malloc(qf_ctx *ctx, size_t bytes, void **dev_out) {
    if (!ctx || !dev_out) return QF_ERR_ARG;
    QF_BIND(ctx);
    QF_HIP(ctx, hipMalloc(dev_out, bytes ? bytes : 16));
    return QF_OK;
}
int qf_free(qf_ctx *ctx, void *dev) {
    if (!ctx) return QF_ERR_ARG;
    QF_BIND(ctx);
    QF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QF_HIP(ctx, hipFree(dev));
    return QF_OK;
}
int qf_memcpy_h2d(qf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!ctx || (!dst_dev && bytes) || (!src_host && bytes)) return QF_ERR_ARG;
    QF_BIND(ctx);
    QF_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QF_OK;
}
int qf_memcpy_d2h(qf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx || (!dst_host && bytes) || (!src_dev && bytes)) return QF_ERR_ARG;
    QF_BIND(ctx);
    QF_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

// ------------------------------------------------------------------ keys
// One device allocation per key: [uint32 copy | digit planes | packed remainder planes], each part
// 256-byte aligned (three hipMalloc/hipFree pairs per key were most of a small key's upload time).
// key_alloc sizes and carves the block; the caller fills k->u32 on ctx->stream; key_finish recodes.
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
// Freed key blocks of up to 16 MiB are parked (at most 8 per process) and handed to the next key of
// exactly the same size on the same device: a host that regenerates small keys in a loop (the
// reference's own bench does, benches/regev.rs:15-57) pays no hipMalloc/hipFree per key.
struct ParkedBlock { int device; size_t bytes; void *ptr; };
static std::mutex g_park_mu;
static std::vector<ParkedBlock> g_parked;
static const size_t PARK_MAX_BYTES = 16u << 20, PARK_MAX_BLOCKS = 8;
static void *park_take(int device, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_park_mu);
    for (size_t i = 0; i < g_parked.size(); i++)
        if (g_parked[i].device == device && g_parked[i].bytes == bytes) {
            void *p = g_parked[i].ptr;
            g_parked.erase(g_parked.begin() + i);
            return p;
        }
    return nullptr;
}
// true if the block was kept (the caller must have synchronised the device first)
static bool park_give(int device, size_t bytes, void *ptr) {
    if (bytes > PARK_MAX_BYTES) return false;
    void *evict = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        if (g_parked.size() >= PARK_MAX_BLOCKS) { evict = g_parked.front().ptr; hipSetDevice(g_parked.front().device); g_parked.erase(g_parked.begin()); }
        g_parked.push_back({device, bytes, ptr});
    }
    if (evict) { hipFree(evict); hipSetDevice(device); }
    return true;
}
static int key_alloc(qf_ctx *ctx, uint32_t rows, uint32_t cols, uint32_t q, qf_key **out) {
    qf_key *k = new (std::nothrow) qf_key();
    if (!k) return qf_fail(ctx, QF_ERR_DEVICE, "out of host memory");
    k->device = ctx->device; k->rows = rows; k->cols = cols; k->q = q;
    k->mod = qf_make_mod(q);
    k->la = qf_digits_for_modulus(q);
    k->rb = (rows + 31) / 32;
    k->ks = ((cols + 127) / 128) * 4;
    const uint32_t r = rows % 32;
    if (r && r * (uint32_t)k->la <= 32) k->rem_rows = r;
    const size_t ubytes = align256((size_t)rows * cols * 4);
    const size_t pbytes = align256((size_t)k->rb * k->ks * k->la * 1024);
    const size_t rbytes = k->rem_rows ? (size_t)k->ks * 1024 : 0;
    k->block_bytes = ubytes + pbytes + rbytes;
    k->block = park_take(ctx->device, k->block_bytes);
    if (!k->block && hipMalloc((void **)&k->block, k->block_bytes) != hipSuccess) {
        delete k;
        return qf_fail(ctx, QF_ERR_DEVICE, "hipMalloc(key) failed");
    }
    k->u32 = (uint32_t *)k->block;
    k->planes = (int8_t *)k->block + ubytes;
    k->rem_planes = k->rem_rows ? (int8_t *)k->block + ubytes + pbytes : nullptr;
    *out = k;
    return QF_OK;
}
static int key_finish(qf_ctx *ctx, qf_key *k, qf_key **out) {
    int rc = qf_launch_recode(ctx, k->u32, QF_X_U32, (size_t)k->cols * 4, k->rows, k->cols, k->mod, k->la, k->ks, k->planes, ctx->flag);
    if (!rc && k->rem_rows)
        rc = qf_launch_recode_rem(ctx, k->u32, k->cols, k->rows - k->rem_rows, k->rem_rows, k->mod, k->la, k->ks, k->rem_planes);
    if (rc) { qf_key_destroy(k); return rc; }
    uint32_t flag = 0;
    if (hipMemcpyAsync(&flag, ctx->flag, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { qf_key_destroy(k); return qf_fail(ctx, QF_ERR_DEVICE, "key recode failed"); }
    if (flag) { hipMemsetAsync(ctx->flag, 0, 4, ctx->stream); qf_key_destroy(k); return qf_fail(ctx, QF_ERR_RANGE, "key recode: digit overflow (internal)"); }
    *out = k;
    return QF_OK;
}

int qf_key_upload_dev(qf_ctx *ctx, const uint32_t *A_dev, uint32_t rows, uint32_t cols, uint32_t q,
                      int transpose, qf_key **out) {
    if (!ctx || !A_dev || !out) return QF_ERR_ARG;
    *out = nullptr;
    if (!rows || !cols) return qf_fail(ctx, QF_ERR_DIM, "key with zero rows or columns");
    int rc = check_q(ctx, q);
    if (rc) return rc;
    QF_BIND(ctx);
    qf_key *k = nullptr;
    rc = transpose ? key_alloc(ctx, cols, rows, q, &k) : key_alloc(ctx, rows, cols, q, &k);
    if (rc) return rc;
    if (transpose) {
        rc = qf_launch_transpose_u32(ctx, A_dev, rows, cols, k->u32);
        if (rc) { qf_key_destroy(k); return rc; }
    } else if (hipMemcpyAsync(k->u32, A_dev, (size_t)rows * cols * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        qf_key_destroy(k);
        return qf_fail(ctx, QF_ERR_DEVICE, "key copy failed");
    }
    return key_finish(ctx, k, out);
}
int qf_key_upload_host(qf_ctx *ctx, const uint32_t *A_host, uint32_t rows, uint32_t cols, uint32_t q,
                       int transpose, qf_key **out) {
    if (!ctx || !A_host || !out) return QF_ERR_ARG;
    *out = nullptr;
    if (!rows || !cols) return qf_fail(ctx, QF_ERR_DIM, "key with zero rows or columns");
    int rc = check_q(ctx, q);
    if (rc) return rc;
    QF_BIND(ctx);
    const size_t bytes = (size_t)rows * cols * 4;
    if (!transpose) {   // H2D straight into the key's own copy
        qf_key *k = nullptr;
        rc = key_alloc(ctx, rows, cols, q, &k);
        if (rc) return rc;
        if (hipMemcpyAsync(k->u32, A_host, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            qf_key_destroy(k);
            return qf_fail(ctx, QF_ERR_DEVICE, "key H2D copy failed");
        }
        return key_finish(ctx, k, out);   // synchronises: the host buffer is the caller's again on return
    }
    uint32_t *tmp = nullptr;
    QF_HIP(ctx, hipMalloc((void **)&tmp, bytes));
    hipError_t e = hipMemcpyAsync(tmp, A_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(tmp); ctx->err = "key H2D copy failed"; return QF_ERR_DEVICE; }
    rc = qf_key_upload_dev(ctx, tmp, rows, cols, q, transpose, out);
    hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    return rc;
}
void qf_key_destroy(qf_key *key) {
    if (!key) return;
    hipSetDevice(key->device);
    hipDeviceSynchronize();   // launches that read the key may still be in flight on any stream
    for (int8_t *f : key->fold_planes) if (f) hipFree(f);
    if (key->block && !park_give(key->device, key->block_bytes, key->block)) hipFree(key->block);
    delete key;
}
int qf_key_dims(const qf_key *key, uint32_t *rows, uint32_t *cols, uint32_t *q) {
    if (!key) return QF_ERR_ARG;
    if (rows) *rows = key->rows;
    if (cols) *cols = key->cols;
    if (q) *q = key->q;
    return QF_OK;
}
// One process driving several GPUs: the uint32 matrix goes from the source key's device to each
// target context's device (peer copy over xGMI; a plain device copy when they share a device), the
// digit planes are derived locally on every GPU -- the same "replicate once, recode locally" step
// that one-process-per-GPU hosts do with qf_key_u32_dev + an RCCL broadcast + qf_key_upload_dev.
int qf_key_replicate(qf_ctx *const *ctxs, int n, const qf_key *src, qf_key **out) {
    if (!ctxs || n <= 0 || !src || !out) return QF_ERR_ARG;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    for (int i = 0; i < n; i++) if (!ctxs[i]) return QF_ERR_ARG;
    const size_t bytes = (size_t)src->rows * src->cols * 4;
    for (int i = 0; i < n; i++) {
        qf_ctx *ctx = ctxs[i];
        int rc = QF_OK;
        qf_key *k = nullptr;
        if (hipSetDevice(ctx->device) != hipSuccess) rc = qf_fail(ctx, QF_ERR_DEVICE, "key replicate: cannot select the target device");
        else rc = key_alloc(ctx, src->rows, src->cols, src->q, &k);
        if (!rc) {
            hipError_t e = ctx->device == src->device
                               ? hipMemcpyAsync(k->u32, src->u32, bytes, hipMemcpyDeviceToDevice, ctx->stream)
                               : hipMemcpyPeerAsync(k->u32, ctx->device, src->u32, src->device, bytes, ctx->stream);
            if (e != hipSuccess) { qf_key_destroy(k); rc = qf_fail(ctx, QF_ERR_DEVICE, "key replicate: device-to-device copy failed"); }
            else rc = key_finish(ctx, k, &out[i]);
        }
        if (rc) {
            for (int j = 0; j < i; j++) { qf_key_destroy(out[j]); out[j] = nullptr; }
            return rc;
        }
    }
    return QF_OK;
}
int qf_key_u32_dev(const qf_key *key, const uint32_t **dev_ptr, size_t *bytes) {
    if (!key || !dev_ptr || !bytes) return QF_ERR_ARG;
    *dev_ptr = key->u32;
    *bytes = (size_t)key->rows * key->cols * 4;
    return QF_OK;
}

// ------------------------------------------------------------------ bit rows -> k-tile-major planes
// One workgroup moves 64 rows x 8 k-tiles (one 128-byte line per row): read 8 threads per row, written 64 rows per
// k-tile (1 KiB runs), the transposition goes through LDS.
__global__ __launch_bounds__(512) void qf_bits_to_kt_kernel(const uint8_t *__restrict__ in, size_t stride, uint64_t B, uint32_t ntiles,
                                                             uint8_t *__restrict__ out, size_t kt_stride) {
    __shared__ uint4 tile[8][65];
    const uint64_t b0 = (uint64_t)blockIdx.x * 64;
    const uint32_t t0 = blockIdx.y * 8;
    {
        const uint32_t r = threadIdx.x >> 3, tt = threadIdx.x & 7;
        uint4 v = {0, 0, 0, 0};
        if (b0 + r < B && t0 + tt < ntiles && (size_t)(t0 + tt) * 16 + 16 <= stride)
            v = *(const uint4 *)(in + (b0 + r) * stride + (size_t)(t0 + tt) * 16);
        tile[tt][r] = v;
    }
    __syncthreads();
    const uint32_t tt = threadIdx.x >> 6, r = threadIdx.x & 63;
    if (b0 + r < B && t0 + tt < ntiles) *(uint4 *)(out + (size_t)(t0 + tt) * kt_stride + (b0 + r) * 16) = tile[tt][r];
}
int qf_bits_to_kt_dev(qf_ctx *ctx, const uint8_t *bits_dev, size_t stride_bytes, uint64_t B, uint32_t cols,
                      uint8_t *kt_dev, size_t kt_stride_bytes) {
    if (!ctx) return QF_ERR_ARG;
    if (B == 0 || cols == 0) return QF_OK;
    if (!bits_dev || !kt_dev) return qf_fail(ctx, QF_ERR_ARG, "bits_to_kt: null operand");
    if (stride_bytes % 16 || kt_stride_bytes % 16 || ((uintptr_t)bits_dev & 15) || ((uintptr_t)kt_dev & 15))
        return qf_fail(ctx, QF_ERR_ARG, "bits_to_kt: rows and planes must be 16-byte aligned/strided");
    if (stride_bytes * 8 < cols) return qf_fail(ctx, QF_ERR_DIM, "bits_to_kt: bit rows shorter than cols");
    if (kt_stride_bytes / 16 < B) return qf_fail(ctx, QF_ERR_DIM, "bits_to_kt: planes shorter than the batch");
    const uint32_t ntiles = (cols + 127) / 128;
    const uint64_t bx = (B + 63) / 64;
    if (bx > 0x7fffffffull) return qf_fail(ctx, QF_ERR_NOTSUP, "bits_to_kt: batch too large");
    QF_BIND(ctx);
    hipLaunchKernelGGL(qf_bits_to_kt_kernel, dim3((unsigned)bx, (ntiles + 7) / 8), dim3(512), 0, ctx->stream, bits_dev, stride_bytes, B,
                       ntiles, kt_dev, kt_stride_bytes);
    QF_HIP(ctx, hipGetLastError());
    return QF_OK;
}

// ------------------------------------------------------------------ wide operand -> digit planes, ahead of the product
size_t qf_operand_planes_bytes(const qf_key *key, uint64_t B, int digits) {
    if (!key || digits < 1 || digits > key->la) return 0;
    return (size_t)((B + 31) / 32) * key->ks * (size_t)digits * 1024;
}
int qf_recode_planes_dev(qf_ctx *ctx, const qf_key *key, const void *X_dev, int x_kind, size_t x_stride_bytes, int digits,
                         uint64_t B, int8_t *planes_dev) {
    if (!ctx || !key) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!X_dev || !planes_dev || ((uintptr_t)planes_dev & 15)) return qf_fail(ctx, QF_ERR_ARG, "recode planes: null or misaligned operand");
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "recode planes: key lives on another device");
    if (x_kind != QF_X_U32 && x_kind != QF_X_I32) return qf_fail(ctx, QF_ERR_ARG, "recode planes: operand must be QF_X_U32 or QF_X_I32");
    if (x_stride_bytes < (size_t)key->cols * 4 || x_stride_bytes % 4) return qf_fail(ctx, QF_ERR_DIM, "recode planes: operand rows shorter than the key's column count");
    if (digits < 1 || digits > key->la) return qf_fail(ctx, QF_ERR_ARG, "recode planes: digits must be 1..digits of q");
    if (B > 0xFFFFFFFFull) return qf_fail(ctx, QF_ERR_NOTSUP, "recode planes: batch too large");
    QF_BIND(ctx);
    int rc = qf_launch_recode(ctx, X_dev, x_kind, x_stride_bytes, (uint32_t)B, key->cols, key->mod, digits, key->ks, planes_dev, ctx->flag);
    if (rc) return rc;
    if (digits < key->la) {   // the promise: read the device's verdict before anybody can use the planes
        QF_HIP(ctx, hipMemcpyAsync(ctx->hflag, ctx->flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        QF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (*(volatile uint32_t *)ctx->hflag != 0) {
            QF_HIP(ctx, hipMemsetAsync(ctx->flag, 0, 4, ctx->stream));
            return qf_fail(ctx, QF_ERR_RANGE, "recode planes: an entry does not fit the promised digits");
        }
    }
    return QF_OK;
}

// ------------------------------------------------------------------ kernel (a)
int qf_matmul_modq_dev(qf_ctx *ctx, const qf_key *key, const void *X_dev, int x_kind, size_t x_stride_bytes,
                       int x_digits, uint64_t B, const uint32_t *addend_dev, const uint8_t *msg_dev,
                       uint32_t *out_dev) {
    if (!ctx || !key) return QF_ERR_ARG;
    if (B == 0) return QF_OK;  // empty batch: nothing to do
    if (!X_dev || !out_dev) return qf_fail(ctx, QF_ERR_ARG, "matmul: null operand");
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "matmul: key lives on another device");
    QF_BIND(ctx);
    if (x_kind == QF_X_BITS) {
        if (x_stride_bytes % 16 || ((uintptr_t)X_dev & 15)) return qf_fail(ctx, QF_ERR_ARG, "matmul: bit rows must be 16-byte aligned/strided");
        if (x_stride_bytes * 8 < key->cols) return qf_fail(ctx, QF_ERR_DIM, "matmul: bit rows shorter than the key's column count");
        return qf_launch_gemm(ctx, key, X_dev, 1, x_stride_bytes, 1, B, addend_dev, msg_dev, out_dev);
    }
    if (x_kind == QF_X_BITS_KT) {
        // x_stride_bytes = distance between the planes of two k-tiles; ceil(cols / 128) planes of B 16-byte words
        if (x_stride_bytes % 16 || ((uintptr_t)X_dev & 15)) return qf_fail(ctx, QF_ERR_ARG, "matmul: k-tile planes must be 16-byte aligned/strided");
        if (x_stride_bytes / 16 < B) return qf_fail(ctx, QF_ERR_DIM, "matmul: k-tile planes shorter than the batch");
        return qf_launch_gemm(ctx, key, X_dev, 2, x_stride_bytes, 1, B, addend_dev, msg_dev, out_dev);
    }
    if (x_kind == QF_X_PLANES) {
        if (x_digits < 1 || x_digits > key->la) return qf_fail(ctx, QF_ERR_ARG, "matmul: digit planes need x_digits = their digit count (1..digits of q)");
        if ((uintptr_t)X_dev & 15) return qf_fail(ctx, QF_ERR_ARG, "matmul: digit planes must be 16-byte aligned");
        if (B > 0xFFFFFFFFull) return qf_fail(ctx, QF_ERR_NOTSUP, "matmul: batch too large");
        return qf_launch_gemm(ctx, key, X_dev, 0, 0, x_digits, B, addend_dev, msg_dev, out_dev);
    }
    if (x_kind != QF_X_U32 && x_kind != QF_X_I32) return qf_fail(ctx, QF_ERR_ARG, "matmul: unknown operand kind");
    if (x_stride_bytes < (size_t)key->cols * 4 || x_stride_bytes % 4) return qf_fail(ctx, QF_ERR_DIM, "matmul: operand rows shorter than the key's column count");
    if (B > 0xFFFFFFFFull) return qf_fail(ctx, QF_ERR_NOTSUP, "matmul: batch too large");
    int lb = (x_digits >= 1 && x_digits < key->la) ? x_digits : key->la;
    // A digit promise (lb < la) is verified by the recode kernel; the product is launched right behind it
    // WITHOUT waiting for the verdict -- the host reads the flag (copied to pinned memory behind the recode)
    // while the GEMM runs, and only a broken promise costs anything: the product is redone at full width
    // over the same output (exact for every input).
    for (;;) {
        const bool promised = lb < key->la;
        size_t pbytes = (size_t)((B + 31) / 32) * key->ks * lb * 1024;
        int rc = qf_ws_reserve(ctx, pbytes);
        if (rc) return rc;
        rc = qf_launch_recode(ctx, X_dev, x_kind, x_stride_bytes, (uint32_t)B, key->cols, key->mod, lb, key->ks,
                              (int8_t *)ctx->ws, ctx->flag);
        if (rc) return rc;
        if (promised) {
            QF_HIP(ctx, hipMemcpyAsync(ctx->hflag, ctx->flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            QF_HIP(ctx, hipEventRecord(ctx->evf, ctx->stream));
        }
        // in-place accumulation (addend == out): a product over unverified planes would destroy the addend
        const bool in_place = promised && (const void *)addend_dev == (const void *)out_dev;
        bool broken = false;
        if (in_place) {
            QF_HIP(ctx, hipEventSynchronize(ctx->evf));
            broken = *(volatile uint32_t *)ctx->hflag != 0;
        }
        if (!broken) {
            rc = qf_launch_gemm(ctx, key, ctx->ws, 0, 0, lb, B, addend_dev, msg_dev, out_dev);
            if (rc || !promised) return rc;
            if (!in_place) {
                QF_HIP(ctx, hipEventSynchronize(ctx->evf));
                broken = *(volatile uint32_t *)ctx->hflag != 0;
            }
            if (!broken) return QF_OK;
        }
        QF_HIP(ctx, hipMemsetAsync(ctx->flag, 0, 4, ctx->stream));
        lb = key->la;  // promise broken: redo at full width
    }
}

// ------------------------------------------------------------------ kernel (c)
int qf_matvec_modq_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev) {
    if (!ctx || !key || !x_dev || !y_dev) return QF_ERR_ARG;
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "matvec: key lives on another device");
    QF_BIND(ctx);
    return qf_launch_matvec(ctx, key, x_dev, y_dev);
}
int qf_vecmat_modq_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev) {
    if (!ctx || !key || !x_dev || !y_dev) return QF_ERR_ARG;
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "vecmat: key lives on another device");
    QF_BIND(ctx);
    return qf_launch_vecmat(ctx, key, x_dev, y_dev);
}
int qf_vecmat_u32_dev(qf_ctx *ctx, const uint32_t *A_dev, uint32_t rows, uint32_t cols, uint32_t q,
                      const uint32_t *x_dev, const uint32_t *addend_dev, uint32_t *y_dev) {
    if (!ctx || !A_dev || !x_dev || !y_dev) return QF_ERR_ARG;
    if (!rows || !cols) return qf_fail(ctx, QF_ERR_DIM, "vecmat: matrix with zero rows or columns");
    int rc = check_q(ctx, q);
    if (rc) return rc;
    QF_BIND(ctx);
    return qf_launch_vecmat_raw(ctx, A_dev, rows, cols, qf_make_mod(q), x_dev, addend_dev, y_dev);
}
int qf_matvec_modq_static_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev) {
    if (!ctx || !key || !x_dev || !y_dev) return QF_ERR_ARG;
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "matvec: key lives on another device");
    QF_BIND(ctx);
    return qf_launch_matvec_static(ctx, key, x_dev, y_dev);
}
static int vec_host(qf_ctx *ctx, const qf_key *key, const uint32_t *x_host, uint32_t *y_host, bool trans) {
    if (!ctx || !key || !x_host || !y_host) return QF_ERR_ARG;
    if (key->device != ctx->device) return qf_fail(ctx, QF_ERR_ARG, "matvec: key lives on another device");
    QF_BIND(ctx);
    uint32_t nin = trans ? key->rows : key->cols, nout = trans ? key->cols : key->rows;
    // staging pooled in the context (grown on demand): single-message calls allocate nothing in steady state
    const size_t words = (((size_t)nin + 3) & ~(size_t)3) + nout + 8;
    if (words > ctx->hostvec_words) {
        if (ctx->hostvec) { QF_HIP(ctx, hipStreamSynchronize(ctx->stream)); QF_HIP(ctx, hipFree(ctx->hostvec)); ctx->hostvec = nullptr; ctx->hostvec_words = 0; }
        QF_HIP(ctx, hipMalloc((void **)&ctx->hostvec, words * 4 * 2));
        ctx->hostvec_words = words * 2;
    }
    uint32_t *x = ctx->hostvec, *y = ctx->hostvec + ((nin + 3) & ~3u);
    int rc = QF_OK;
    if (hipMemcpyAsync(x, x_host, (size_t)nin * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = QF_ERR_DEVICE;
    if (!rc) rc = trans ? qf_launch_vecmat(ctx, key, x, y) : qf_launch_matvec(ctx, key, x, y);
    if (!rc && hipMemcpyAsync(y_host, y, (size_t)nout * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = QF_ERR_DEVICE;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = QF_ERR_DEVICE;
    return rc;
}
int qf_matvec_modq_host(qf_ctx *ctx, const qf_key *key, const uint32_t *x_host, uint32_t *y_host) {
    return vec_host(ctx, key, x_host, y_host, false);
}
int qf_vecmat_modq_host(qf_ctx *ctx, const qf_key *key, const uint32_t *x_host, uint32_t *y_host) {
    return vec_host(ctx, key, x_host, y_host, true);
}

int qf_dot_decode_dev(qf_ctx *ctx, const uint32_t *w_dev, uint32_t n, uint32_t q, const uint32_t *C_dev,
                      uint64_t B, uint8_t *bits_dev) {
    if (!ctx) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!w_dev || !C_dev || !bits_dev) return qf_fail(ctx, QF_ERR_ARG, "decode: null operand");
    int rc = check_q(ctx, q);
    if (rc) return rc;
    QF_BIND(ctx);
    return qf_launch_dot_decode(ctx, w_dev, n, qf_make_mod(q), C_dev, B, bits_dev);
}

// ------------------------------------------------------------------ rings
int qf_ring_create(qf_ctx *ctx, uint32_t n, uint32_t q, qf_ring **out) {
    if (!ctx || !out) return QF_ERR_ARG;
    *out = nullptr;
    int rc = check_q(ctx, q);
    if (rc) return rc;
    if (n < 2 || n > 4096 || (n & (n - 1))) return qf_fail(ctx, QF_ERR_NOTSUP, "ring degree must be a power of two in [2, 4096]");
    QF_BIND(ctx);
    qf_ring *r = new (std::nothrow) qf_ring();
    if (!r) return QF_ERR_DEVICE;
    r->device = ctx->device; r->n = n; r->q = q; r->mod = qf_make_mod(q);
    while ((1u << r->lg) < n) r->lg++;
    r->ntt = 0;
    if (h_is_prime(q) && (q - 1) % (2ull * n) == 0) {
        uint64_t e = (uint64_t)(q - 1) / (2ull * n);
        uint32_t psi = 0;
        for (uint32_t g = 2; g < q && !psi; g++) {
            uint32_t c = (uint32_t)h_powmod(g, e, q);
            if (h_powmod(c, n, q) == q - 1) psi = c;  // order exactly 2n (n is a power of two)
        }
        if (psi) {
            std::vector<uint32_t> tw(4 * (size_t)n);
            uint32_t ipsi = (uint32_t)h_powmod(psi, q - 2, q);
            for (uint32_t i = 0; i < n; i++) {
                uint32_t br = h_bitrev(i, (int)r->lg);
                uint32_t f = (uint32_t)h_powmod(psi, br, q), iv = (uint32_t)h_powmod(ipsi, br, q);
                tw[i] = f; tw[n + i] = (uint32_t)((((uint64_t)f) << 32) / q);
                tw[2 * n + i] = iv; tw[3 * n + i] = (uint32_t)((((uint64_t)iv) << 32) / q);
            }
            r->ninv = (uint32_t)h_powmod(n % q, q - 2, q);
            r->ninv_p = (uint32_t)((((uint64_t)r->ninv) << 32) / q);
            if (hipMalloc((void **)&r->tw, tw.size() * 4) != hipSuccess ||
                hipMemcpyAsync(r->tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                qf_ring_destroy(r);
                return qf_fail(ctx, QF_ERR_DEVICE, "twiddle upload failed");
            }
            r->ntt = 1;
            r->wave = qf_ring_wave_supported(r->lg) ? 1 : 0;
        }
    }
    *out = r;
    return QF_OK;
}
void qf_ring_destroy(qf_ring *ring) {
    if (!ring) return;
    hipSetDevice(ring->device);
    hipDeviceSynchronize();
    if (ring->tw) hipFree(ring->tw);
    delete ring;
}
int qf_ring_info(const qf_ring *ring, uint32_t *n, uint32_t *q, int *uses_ntt) {
    if (!ring) return QF_ERR_ARG;
    if (n) *n = ring->n;
    if (q) *q = ring->q;
    if (uses_ntt) *uses_ntt = ring->ntt;
    return QF_OK;
}

int qf_rkey_upload_dev(qf_ctx *ctx, const qf_ring *ring, const uint32_t *polys_dev, uint32_t npoly, qf_rkey **out) {
    if (!ctx || !ring || !polys_dev || !out) return QF_ERR_ARG;
    *out = nullptr;
    if (!npoly) return qf_fail(ctx, QF_ERR_DIM, "ring key without polynomials");
    QF_BIND(ctx);
    qf_rkey *k = new (std::nothrow) qf_rkey();
    if (!k) return QF_ERR_DEVICE;
    k->device = ctx->device; k->n = ring->n; k->q = ring->q; k->npoly = npoly; k->ntt = ring->ntt; k->wave = ring->wave;
    size_t bytes = (size_t)npoly * ring->n * 4;
    if (hipMalloc((void **)&k->coef, bytes) != hipSuccess) { delete k; return qf_fail(ctx, QF_ERR_DEVICE, "hipMalloc(ring key) failed"); }
    // canonicalise through the sampler-free path: copy, then (ntt rings) transform
    if (hipMemcpyAsync(k->coef, polys_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { qf_rkey_destroy(k); return qf_fail(ctx, QF_ERR_DEVICE, "ring key copy failed"); }
    if (ring->ntt) {
        if (hipMalloc((void **)&k->hat, ring->wave ? 2 * bytes : bytes) != hipSuccess ||
            (!ring->wave && hipMalloc((void **)&k->hat_p, bytes) != hipSuccess)) { qf_rkey_destroy(k); return qf_fail(ctx, QF_ERR_DEVICE, "hipMalloc(ring key) failed"); }
        int rc = qf_launch_ntt_forward_keys(ctx, ring, k);
        if (rc) { qf_rkey_destroy(k); return rc; }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { qf_rkey_destroy(k); return qf_fail(ctx, QF_ERR_DEVICE, "ring key upload failed"); }
    *out = k;
    return QF_OK;
}
int qf_rkey_upload_host(qf_ctx *ctx, const qf_ring *ring, const uint32_t *polys_host, uint32_t npoly, qf_rkey **out) {
    if (!ctx || !ring || !polys_host || !out) return QF_ERR_ARG;
    *out = nullptr;
    if (!npoly) return qf_fail(ctx, QF_ERR_DIM, "ring key without polynomials");
    QF_BIND(ctx);
    size_t bytes = (size_t)npoly * ring->n * 4;
    std::vector<uint32_t> canon(polys_host, polys_host + (size_t)npoly * ring->n);
    for (auto &c : canon) c %= ring->q;
    uint32_t *tmp = nullptr;
    QF_HIP(ctx, hipMalloc((void **)&tmp, bytes));
    hipError_t e = hipMemcpyAsync(tmp, canon.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(tmp); return qf_fail(ctx, QF_ERR_DEVICE, "ring key H2D failed"); }
    int rc = qf_rkey_upload_dev(ctx, ring, tmp, npoly, out);
    hipFree(tmp);
    return rc;
}
void qf_rkey_destroy(qf_rkey *k) {
    if (!k) return;
    hipSetDevice(k->device);
    hipDeviceSynchronize();
    if (k->coef) hipFree(k->coef);
    if (k->hat) hipFree(k->hat);
    if (k->hat_p) hipFree(k->hat_p);
    delete k;
}

static int ring_match(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *k, uint32_t need) {
    if (k->n != ring->n) return qf_fail(ctx, QF_ERR_DIM, "ring key degree differs from the ring");
    if (k->q != ring->q) return qf_fail(ctx, QF_ERR_MODULUS, "ring key modulus differs from the ring");
    if (k->npoly < need) return qf_fail(ctx, QF_ERR_DIM, "ring key has too few polynomials");
    return QF_OK;
}

int qf_polymul_dev(qf_ctx *ctx, const qf_ring *ring, const uint32_t *a_dev, const uint32_t *b_dev, uint64_t B, uint32_t *c_dev) {
    if (!ctx || !ring) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!a_dev || !b_dev || !c_dev) return qf_fail(ctx, QF_ERR_ARG, "polymul: null operand");
    QF_BIND(ctx);
    return qf_launch_polymul(ctx, ring, a_dev, b_dev, B, c_dev);
}
int qf_ringlpr_enc_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *pk, const uint32_t *r_dev,
                       const uint32_t *e1_dev, const uint32_t *e2_dev, const uint8_t *mu_bits_dev, uint64_t B,
                       uint32_t *u_dev, uint32_t *v_dev) {
    if (!ctx || !ring || !pk) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!r_dev || !e1_dev || !e2_dev || !mu_bits_dev || !u_dev || !v_dev) return qf_fail(ctx, QF_ERR_ARG, "ringlpr enc: null operand");
    int rc = ring_match(ctx, ring, pk, 2);
    if (rc) return rc;
    if (ring->wave && ((uintptr_t)mu_bits_dev & 3)) return qf_fail(ctx, QF_ERR_ARG, "ringlpr enc: mu_bits must be 4-byte aligned");
    QF_BIND(ctx);
    return qf_launch_ringlpr_enc(ctx, ring, pk, r_dev, e1_dev, e2_dev, mu_bits_dev, B, u_dev, v_dev);
}
int qf_ringlpr_dec_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *sk, const uint32_t *u_dev,
                       const uint32_t *v_dev, uint64_t B, uint8_t *mu_bits_dev) {
    if (!ctx || !ring || !sk) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!u_dev || !v_dev || !mu_bits_dev) return qf_fail(ctx, QF_ERR_ARG, "ringlpr dec: null operand");
    int rc = ring_match(ctx, ring, sk, 1);
    if (rc) return rc;
    if (ring->wave && ((uintptr_t)mu_bits_dev & 3)) return qf_fail(ctx, QF_ERR_ARG, "ringlpr dec: mu_bits must be 4-byte aligned");
    QF_BIND(ctx);
    return qf_launch_ringlpr_dec(ctx, ring, sk, u_dev, v_dev, B, mu_bits_dev);
}
int qf_fdh_ring_verify_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *a, const int32_t *sigma_dev,
                           const uint32_t *u_dev, uint64_t norm_bound, uint64_t B, uint32_t *t_dev, uint8_t *ok_dev) {
    if (!ctx || !ring || !a) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!sigma_dev || !u_dev || !ok_dev) return qf_fail(ctx, QF_ERR_ARG, "fdh verify: null operand");
    int rc = ring_match(ctx, ring, a, 1);
    if (rc) return rc;
    QF_BIND(ctx);
    return qf_launch_fdh_verify(ctx, ring, a, sigma_dev, u_dev, norm_bound, B, t_dev, ok_dev);
}

int qf_hash_to_mat_zq_sha256_dev(qf_ctx *ctx, const uint8_t *msgs_dev, const uint64_t *offsets_dev, uint64_t B,
                                 uint32_t rows, uint32_t cols, uint32_t q, uint32_t *out_dev) {
    if (!ctx) return QF_ERR_ARG;
    if (B == 0) return QF_OK;
    if (!offsets_dev || !out_dev) return qf_fail(ctx, QF_ERR_ARG, "hash: null operand");
    if (rows == 0 || cols == 0) return qf_fail(ctx, QF_ERR_DIM, "hash: number of rows and columns must be positive");
    int rc = check_q(ctx, q);
    if (rc) return rc;
    QF_BIND(ctx);
    return qf_launch_hash_to_zq(ctx, msgs_dev, offsets_dev, B, rows, cols, qf_make_mod(q), out_dev);
}

// ------------------------------------------------------------------ generator
// seeded: SplitMix64 under seed ^ (stream << 48); keyed: ChaCha20 under a 256-bit key, nonce ^ (stream << 48)
static QfGen gen_seeded(uint64_t seed, uint32_t stream) {
    QfGen g;
    memset(&g, 0, sizeof g);
    g.base = seed ^ ((uint64_t)stream << 48);
    return g;
}
static QfGen gen_keyed(const uint8_t *key, uint64_t nonce, uint32_t stream) {
    QfGen g = gen_seeded(nonce, stream);
    for (int i = 0; i < 8; i++)
        g.key[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    g.keyed = 1;
    return g;
}
static int sample_uniform(qf_ctx *ctx, const QfGen &g, uint64_t first, uint64_t count, uint32_t q, uint32_t *out_dev) {
    if (!ctx || (!out_dev && count)) return QF_ERR_ARG;
    if (q < 1) return qf_fail(ctx, QF_ERR_MODULUS, "sample: bound must be positive");
    QF_BIND(ctx);
    return qf_launch_sample_uniform(ctx, g, first, count, q, out_dev);
}
static int sample_bits(qf_ctx *ctx, const QfGen &g, uint64_t first_row, uint64_t rows, uint32_t cols, uint32_t stride_bytes, uint8_t *out_dev) {
    if (!ctx || (!out_dev && rows)) return QF_ERR_ARG;
    if (stride_bytes % 4 || (uint64_t)stride_bytes * 8 < cols) return qf_fail(ctx, QF_ERR_DIM, "sample bits: stride must be a multiple of 4 bytes covering cols bits");
    QF_BIND(ctx);
    return qf_launch_sample_bits(ctx, g, first_row, rows, cols, stride_bytes, out_dev);
}
static int sample_bits_kt(qf_ctx *ctx, const QfGen &g, uint64_t first_row, uint64_t rows, uint32_t cols, size_t kt_stride_bytes, uint8_t *out_dev) {
    if (!ctx || (!out_dev && rows && cols)) return QF_ERR_ARG;
    if (kt_stride_bytes % 16 || ((uintptr_t)out_dev & 15)) return qf_fail(ctx, QF_ERR_ARG, "sample bits (k-tile-major): planes must be 16-byte aligned/strided");
    if (kt_stride_bytes / 16 < rows) return qf_fail(ctx, QF_ERR_DIM, "sample bits (k-tile-major): planes shorter than the rows");
    if ((cols + 127) / 128 > 65535) return qf_fail(ctx, QF_ERR_NOTSUP, "sample bits (k-tile-major): too many columns");
    QF_BIND(ctx);
    return qf_launch_sample_bits_kt(ctx, g, first_row, rows, cols, kt_stride_bytes, out_dev);
}
static int sample_noise(qf_ctx *ctx, const QfGen &g, uint64_t first, uint64_t count, uint32_t q, uint32_t W, uint32_t *out_dev) {
    if (!ctx || (!out_dev && count)) return QF_ERR_ARG;
    int rc = check_q(ctx, q);
    if (rc) return rc;
    if (W >= (1u << 30)) return qf_fail(ctx, QF_ERR_ARG, "sample noise: W too large");
    QF_BIND(ctx);
    return qf_launch_sample_noise(ctx, g, first, count, qf_make_mod(q), W, out_dev);
}
int qf_sample_uniform_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t *out_dev) {
    return sample_uniform(ctx, gen_seeded(seed, stream), first, count, q, out_dev);
}
int qf_sample_bits_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first_row, uint64_t rows, uint32_t cols,
                       uint32_t stride_bytes, uint8_t *out_dev) {
    return sample_bits(ctx, gen_seeded(seed, stream), first_row, rows, cols, stride_bytes, out_dev);
}
int qf_sample_bits_kt_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first_row, uint64_t rows, uint32_t cols,
                          size_t kt_stride_bytes, uint8_t *out_dev) {
    return sample_bits_kt(ctx, gen_seeded(seed, stream), first_row, rows, cols, kt_stride_bytes, out_dev);
}
int qf_sample_noise_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t W, uint32_t *out_dev) {
    return sample_noise(ctx, gen_seeded(seed, stream), first, count, q, W, out_dev);
}
int qf_sample_uniform_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first, uint64_t count,
                                uint32_t q, uint32_t *out_dev) {
    if (!key32) return QF_ERR_ARG;
    return sample_uniform(ctx, gen_keyed(key32, nonce, stream), first, count, q, out_dev);
}
int qf_sample_bits_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first_row, uint64_t rows,
                             uint32_t cols, uint32_t stride_bytes, uint8_t *out_dev) {
    if (!key32) return QF_ERR_ARG;
    return sample_bits(ctx, gen_keyed(key32, nonce, stream), first_row, rows, cols, stride_bytes, out_dev);
}
int qf_sample_bits_kt_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first_row, uint64_t rows,
                                uint32_t cols, size_t kt_stride_bytes, uint8_t *out_dev) {
    if (!key32) return QF_ERR_ARG;
    return sample_bits_kt(ctx, gen_keyed(key32, nonce, stream), first_row, rows, cols, kt_stride_bytes, out_dev);
}
int qf_sample_noise_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first, uint64_t count,
                              uint32_t q, uint32_t W, uint32_t *out_dev) {
    if (!key32) return QF_ERR_ARG;
    return sample_noise(ctx, gen_keyed(key32, nonce, stream), first, count, q, W, out_dev);
}

// D_{Z,s}: rho_s(x) = exp(-pi x^2 / s^2) on |x| <= T = ceil(s * log2(n)) (GPV08 SampleZ's support with
// t = log2 n), as thresholds floor(2^64 * cumulative / total); sums run in long double from -T up.
static int build_cdt(qf_ctx *ctx, uint32_t n, double s) {
    if (ctx->cdt && ctx->cdt_n == n && ctx->cdt_s == s) return QF_OK;
    const long double sl = (long double)s;
    const long double Tl = ceill(sl * log2l((long double)n));
    if (!(Tl >= 1.0L) || Tl > (long double)(1 << 21)) return qf_fail(ctx, QF_ERR_NOTSUP, "discrete gauss: support s*log2(n) must be in [1, 2^21]");
    const int64_t T = (int64_t)Tl;
    const size_t len = (size_t)(2 * T + 1);
    std::vector<long double> cum(len);
    const long double pi = 3.14159265358979323846264338327950288L;
    long double acc = 0.0L;
    for (int64_t x = -T; x <= T; x++) {
        acc += expl(-pi * (long double)x * (long double)x / (sl * sl));
        cum[(size_t)(x + T)] = acc;
    }
    std::vector<uint64_t> thr(len);
    for (size_t i = 0; i < len; i++) {
        const long double f = cum[i] / acc * 18446744073709551616.0L;
        thr[i] = f >= 18446744073709551615.0L ? 0xFFFFFFFFFFFFFFFFull : (uint64_t)f;
    }
    thr[len - 1] = 0xFFFFFFFFFFFFFFFFull;
    QF_HIP(ctx, hipStreamSynchronize(ctx->stream));   // a sampler launch may still read the old table
    if (ctx->cdt) { hipFree(ctx->cdt); ctx->cdt = nullptr; }
    QF_HIP(ctx, hipMalloc((void **)&ctx->cdt, len * 8));
    QF_HIP(ctx, hipMemcpy(ctx->cdt, thr.data(), len * 8, hipMemcpyHostToDevice));
    ctx->cdt_len = (uint32_t)len; ctx->cdt_n = n; ctx->cdt_s = s;
    return QF_OK;
}
static int sample_gauss(qf_ctx *ctx, const QfGen &g, uint64_t first, uint64_t count, uint32_t q, uint32_t n, double s, uint32_t *out_dev) {
    if (!ctx || (!out_dev && count)) return QF_ERR_ARG;
    int rc = check_q(ctx, q);
    if (rc) return rc;
    if (n < 2 || !(s > 0.0)) return qf_fail(ctx, QF_ERR_ARG, "discrete gauss: need n >= 2 and s > 0");
    QF_BIND(ctx);
    rc = build_cdt(ctx, n, s);
    if (rc) return rc;
    return qf_launch_sample_cdt(ctx, g, first, count, qf_make_mod(q), ctx->cdt, ctx->cdt_len, -(int64_t)(ctx->cdt_len / 2), out_dev);
}
int qf_sample_discrete_gauss_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q,
                                 uint32_t n, double s, uint32_t *out_dev) {
    return sample_gauss(ctx, gen_seeded(seed, stream), first, count, q, n, s, out_dev);
}
int qf_sample_discrete_gauss_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first,
                                       uint64_t count, uint32_t q, uint32_t n, double s, uint32_t *out_dev) {
    if (!key32) return QF_ERR_ARG;
    return sample_gauss(ctx, gen_keyed(key32, nonce, stream), first, count, q, n, s, out_dev);
}

// ------------------------------------------------------------------ timing
int qf_set_timing(qf_ctx *ctx, int enabled) {
    if (!ctx) return QF_ERR_ARG;
    ctx->timing = enabled;
    return QF_OK;
}
int qf_last_kernel_ms(qf_ctx *ctx, const char *which, float *ms) {
    if (!ctx || !which || !ms) return QF_ERR_ARG;
    if (!strcmp(which, "gemm")) *ms = ctx->ms_gemm;
    else if (!strcmp(which, "ntt")) *ms = ctx->ms_ntt;
    else if (!strcmp(which, "matvec")) *ms = ctx->ms_matvec;
    else if (!strcmp(which, "decode")) *ms = ctx->ms_decode;
    else return QF_ERR_ARG;
    return QF_OK;
}

}  // extern "C"
