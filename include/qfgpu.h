/*
 * qfgpu.h -- C ABI of the MI355X (gfx950) implementation of qFALL-crypto's data-parallel
 * lattice arithmetic.  This is the drop-in boundary: the reference has no FFI of its own
 * (it calls qfall-math operator overloads, Cargo.toml:10), so every entry point below names
 * the reference call site(s) it replaces (paths relative to /root/reference/src).  The Rust
 * binding a maintainer would add on top of this header is shown in INTEGRATION.md.
 *
 * Conventions
 *   - q < 2^31; Z_q entries are canonical residues in [0,q) stored as uint32_t.
 *   - Matrices are row-major.  Batches are message-major: X[b][cols], C[b][rows].
 *   - Polynomials of R_q = Z_q[X]/(X^n+1) are n coefficients, degree-ascending.
 *   - Bit vectors are LSB-first inside each byte.
 *   - Pointers named *_dev are DEVICE pointers (hipMalloc / qf_malloc / torch .data_ptr());
 *     pointers named *_host are host pointers.  Nothing is retained after a call returns
 *     except inside the opaque handles.
 *   - Every function returns QF_OK (0) or a QF_ERR_* code; qf_last_error(ctx) gives a text.
 *     The reference panics where we return nonzero (bad dimension / modulus mismatch:
 *     construction/hash/sis.rs:157-165); a binding turns codes back into panics.
 *   - A context is bound to one device and one stream and is not re-entrant; work is
 *     enqueued on the context's stream.  *_dev entry points are asynchronous w.r.t. the
 *     host unless stated; *_host entry points synchronise before returning.
 *   - Every entry point makes the device of its context (destroy calls: of the handle) current on
 *     the calling thread and leaves it current; a host that drives several GPUs from one thread
 *     re-selects its own device after a call if it needs to.  Kernel attributes are set up per
 *     device, so contexts on different devices of one process are independent.
 *   - A key handle may be shared by contexts of the same device.  The first product with an
 *     L-digit batch operand (L >= 2) attaches a second, L times larger set of digit planes to it
 *     (one allocation, kept until qf_key_destroy; built under a lock, so two contexts that meet
 *     there are safe); a key must not be destroyed while any context still uses it.
 */
#ifndef QFGPU_H
#define QFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    QF_OK = 0,
    QF_ERR_ARG = 1,      /* null pointer / zero size / bad enum */
    QF_ERR_DIM = 2,      /* dimension mismatch (reference: panic "mismatching dimensions") */
    QF_ERR_MODULUS = 3,  /* q < 2, q >= 2^31, or modulus mismatch between operands */
    QF_ERR_DEVICE = 4,   /* HIP runtime error */
    QF_ERR_RANGE = 5,    /* an operand exceeded the range promised by the caller */
    QF_ERR_NOTSUP = 6    /* parameter set outside what the kernels support */
};

/* how a batch operand X is encoded */
enum {
    QF_X_BITS = 0, /* {0,1}^cols bit-packed; row stride multiple of 16 bytes          */
    QF_X_U32 = 1,  /* residues mod q as uint32_t (any value < 2^32 is reduced mod q)  */
    QF_X_I32 = 2,  /* signed integers as int32_t (reduced mod q)                      */
    QF_X_PLANES = 4, /* signed radix-2^8 digit planes in MFMA fragment order, as qf_recode_planes_dev writes them for
                        THIS key: x_digits (1..digits of q) planes per k-step; x_stride_bytes is ignored.  The form the
                        product consumes -- qf_matmul_modq_dev recodes QF_X_U32 / QF_X_I32 operands into it on
                        every call (LPR cfg4: 0.25 of 2.2 ms) */
    QF_X_BITS_KT = 3 /* {0,1}^cols bit-packed, k-tile-major: the 16 bytes that hold columns 128t..128t+127 of row b
                        sit at byte t * x_stride_bytes + 16 b (ceil(cols/128) planes of >= B words; what
                        qf_bits_to_kt_dev writes).  Same product as QF_X_BITS, 3.5-4 % faster on the BASELINE
                        shapes: a wave's 32 rows are 512 contiguous bytes instead of 32 different cache lines */
};

typedef struct qf_ctx qf_ctx;   /* device + stream + scratch                            */
typedef struct qf_key qf_key;   /* a Z_q matrix resident on the device (MatZq operand)  */
typedef struct qf_ring qf_ring; /* R_q = Z_q[X]/(X^n+1): n, q, twiddles                 */
typedef struct qf_rkey qf_rkey; /* fixed polynomials of R_q prepared for multiplication */

/* ------------------------------------------------------------------ context */
const char *qf_version(void);
int qf_device_count(int *count);
/* stream: a hipStream_t (may be NULL = the device's default stream) */
int qf_ctx_create(int device, void *stream, qf_ctx **out);
int qf_ctx_set_stream(qf_ctx *ctx, void *stream);
void qf_ctx_destroy(qf_ctx *ctx);
int qf_sync(qf_ctx *ctx);
const char *qf_last_error(const qf_ctx *ctx);
/* plain device-memory helpers so a host language without HIP bindings can stage buffers */
int qf_malloc(qf_ctx *ctx, size_t bytes, void **dev_out);
int qf_free(qf_ctx *ctx, void *dev);
int qf_memcpy_h2d(qf_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int qf_memcpy_d2h(qf_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ------------------------------------------------------------------ Z_q matrix keys
 * A qf_key is what the reference holds as a MatZq public key / SIS key:
 *   Regev/LPR pk (n+1) x m      construction/pk_encryption/regev.rs:328-330, lpr.rs:316-318
 *   DualRegev pk n x (m+1)      dual_regev.rs:327-329   (upload with transpose=1: enc is s^t*pk)
 *   SISHash key n x m           construction/hash/sis.rs:52-54
 * Upload recodes the matrix into signed radix-2^8 digit planes laid out in int8-MFMA
 * fragment order (DESIGN.md "HBM layout") and keeps a uint32 copy for the single-message
 * mat-vec path.  transpose != 0 stores the transpose of the given matrix. */
int qf_key_upload_host(qf_ctx *ctx, const uint32_t *A_host, uint32_t rows, uint32_t cols,
                       uint32_t q, int transpose, qf_key **out);
int qf_key_upload_dev(qf_ctx *ctx, const uint32_t *A_dev, uint32_t rows, uint32_t cols,
                      uint32_t q, int transpose, qf_key **out);
void qf_key_destroy(qf_key *key);
int qf_key_dims(const qf_key *key, uint32_t *rows, uint32_t *cols, uint32_t *q);
/* device pointer + size of the row-major uint32 copy (rows x cols of the key as stored):
 * this is the buffer a multi-GPU host broadcasts with RCCL before qf_key_upload_dev. */
int qf_key_u32_dev(const qf_key *key, const uint32_t **dev_ptr, size_t *bytes);
/* Multi-GPU step of the path (the reference is single-device and has no counterpart):
 * replicate a key that lives on one GPU to n contexts (normally one per GPU of the node) in ONE call.
 * The uint32 matrix crosses xGMI once per target (peer copy enqueued on the target context's stream);
 * every GPU derives its digit planes locally.  out_keys[i] belongs to ctxs[i]'s device; on error
 * nothing is left allocated.  Hosts that run one PROCESS per GPU broadcast the buffer of
 * qf_key_u32_dev with RCCL and call qf_key_upload_dev on each rank instead (bench.py does). */
int qf_key_replicate(qf_ctx *const *ctxs, int n, const qf_key *src_key, qf_key **out_keys);

/* Kernel (a): batched product on int8 MFMA.
 *   out[b][i] = sum_j key[i][j]*X[b][j]  (+ addend[b][i])  (+ (msg[b]&1)*floor(q/2) if i==rows-1)   mod q
 * Replaces, for b = 0..B-1 at once:
 *   Regev::enc      pk * vec_x, message into last entry        regev.rs:404-411   (X bits, msg)
 *   LPR::enc        pk * vec_r + vec_e, message                lpr.rs:420-427     (X u32, addend, msg)
 *   DualRegev::enc  (vec_s_t * pk + vec_e_t)^t, message        dual_regev.rs:403-410 (key uploaded transposed)
 *   SISHash::hash   &self.key * value                          sis.rs:161-167     (no addend/msg)
 *   PSFGPV::f_a     a * sigma                                  primitive/psf/gpv.rs:189-192
 * x_stride_bytes: distance between consecutive rows of X.  x_digits: 0 = full width, or a
 * promise that every centred entry of X fits in that many signed 8-bit digits (1..4); a broken
 * promise is detected on the device and the product is redone at full width (never wrong).
 * addend_dev / msg_dev may be NULL; addend_dev may BE out_dev (in-place accumulation), any other overlap
 * of the two is undefined.  The call returns with the product queued on the stream; with a digit promise
 * it has waited for the recode pass (not for the product) to learn whether the promise held.  Limits (QF_ERR_NOTSUP beyond them): digits(X) * cols * 2^14 < 2^31
 * (int32 accumulators; 43,690 columns for a 3-digit operand, 131,071 for bits or one digit), fewer than
 * 2^24 key rows, B < 2^32.  out_dev needs exactly B * rows words: nothing beyond row B-1 is written. */
int qf_matmul_modq_dev(qf_ctx *ctx, const qf_key *key, const void *X_dev, int x_kind,
                       size_t x_stride_bytes, int x_digits, uint64_t B,
                       const uint32_t *addend_dev, const uint8_t *msg_dev, uint32_t *out_dev);

/* Bit rows (QF_X_BITS: row b at b * stride_bytes) -> k-tile-major planes (QF_X_BITS_KT: plane t at t * kt_stride_bytes,
 * row b's word at + 16 b); ceil(cols/128) planes are written, kt_stride_bytes >= 16 B.  A layout step of ours in front of
 * kernel (a) -- the reference has no bit-packed form at all (it multiplies by a MatZq of 0/1 entries: regev.rs:401-404,
 * sis.rs:161-167); a producer that writes the planes itself does not need it. */
int qf_bits_to_kt_dev(qf_ctx *ctx, const uint8_t *bits_dev, size_t stride_bytes, uint64_t B, uint32_t cols,
                      uint8_t *kt_dev, size_t kt_stride_bytes);

/* A wide batch operand (QF_X_U32 / QF_X_I32 rows) -> the digit planes kernel (a) consumes (QF_X_PLANES), once, for
 * operands that are multiplied more than once or prepared ahead of the product (LPR::enc's r: lpr.rs:399-420).
 * digits: 1..digits of q; fewer than q's is a promise that every centred entry fits -- verified here, synchronously:
 * QF_ERR_RANGE if it does not hold (the planes are then unusable).  planes_dev: qf_operand_planes_bytes(key, B, digits)
 * bytes (= ceil(B/32) * 4*ceil(cols/128) * digits KiB), 16-byte aligned. */
size_t qf_operand_planes_bytes(const qf_key *key, uint64_t B, int digits);
int qf_recode_planes_dev(qf_ctx *ctx, const qf_key *key, const void *X_dev, int x_kind, size_t x_stride_bytes,
                         int digits, uint64_t B, int8_t *planes_dev);

/* Kernel (c): HBM-streaming single-message products (host vectors in/out, synchronous).
 *   qf_matvec_modq_host : y = key * x          regev.rs:404, lpr.rs:420, sis.rs:166, dual_regev.rs:354
 *   qf_vecmat_modq_host : y^t = x^t * key      regev.rs:367, lpr.rs:363, dual_regev.rs:403
 * x has cols (resp. rows) entries (any uint32_t value is reduced mod q), y rows (resp. cols).
 * The mat-vec streams the key's int8 digit planes (ceil(bits(q)/8) bytes per entry); its
 * cross-workgroup sums use a small scratch owned by the context (allocated on first use). */
int qf_matvec_modq_host(qf_ctx *ctx, const qf_key *key, const uint32_t *x_host, uint32_t *y_host);
int qf_vecmat_modq_host(qf_ctx *ctx, const qf_key *key, const uint32_t *x_host, uint32_t *y_host);
/* device-pointer forms (asynchronous) used by the benchmarks */
int qf_matvec_modq_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev);
int qf_vecmat_modq_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev);
/* y^t = x^t*A + addend^t (mod q) for a plain row-major rows x cols uint32 matrix in device memory --
 * no key object, no digit planes: the one-off products of key generation, where A is used once as
 * it comes out of the sampler: b^t = s^t*A + e^t (regev.rs:366-368), the rows of lpr.rs:362-364.
 * A, x and addend may hold any uint32 values (reduced exactly); addend_dev may be NULL. */
int qf_vecmat_u32_dev(qf_ctx *ctx, const uint32_t *A_dev, uint32_t rows, uint32_t cols, uint32_t q,
                      const uint32_t *x_dev, const uint32_t *addend_dev, uint32_t *y_dev);
/* qf_matvec_modq_dev for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * result as qf_matvec_modq_dev; the cross-workgroup sums live in a file-scope device array of the
 * library (256 KiB per GPU, held until the process ends) instead of context-owned scratch, and the
 * call performs no hipMalloc/hipFree.  Limits: key rows (padded to a multiple of 32) <= 32768, else
 * QF_ERR_NOTSUP.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two
 * streams, two contexts, two processes' worth of threads) race -- serialise them. */
int qf_matvec_modq_static_dev(qf_ctx *ctx, const qf_key *key, const uint32_t *x_dev, uint32_t *y_dev);

/* Batched decryption:  r_b = <w, C[b][0..n)> + C[b][n] mod q ;
 *   bit_b = [dist(r_b,0) > dist(r_b, floor(q/2))]  (ring distance, ties -> 0)
 * with w = -sk mod q supplied by the caller as canonical residues:
 *   Regev::dec / LPR::dec   [-s ; 1] . c            regev.rs:438-452, lpr.rs:454-468
 *   DualRegev::dec          c^t * [-x ; 1]          dual_regev.rs:437-450
 * C is B x (n+1); bits_dev receives B bytes (0/1). */
int qf_dot_decode_dev(qf_ctx *ctx, const uint32_t *w_dev, uint32_t n, uint32_t q,
                      const uint32_t *C_dev, uint64_t B, uint8_t *bits_dev);

/* ------------------------------------------------------------------ R_q
 * qf_ring_create replaces utils/common_moduli.rs:41-48 (new_anticyclic): the ring X^n+1 mod q.
 * n must be a power of two, 2 <= n <= 4096.  If q is prime with 2n | q-1 products use the
 * negacyclic NTT (kernel b); for every other q they use an exact schoolbook kernel, so the
 * reference's own parameter sets (ring_lpr.rs:312-314,343-345) work unchanged. */
int qf_ring_create(qf_ctx *ctx, uint32_t n, uint32_t q, qf_ring **out);
void qf_ring_destroy(qf_ring *ring);
int qf_ring_info(const qf_ring *ring, uint32_t *n, uint32_t *q, int *uses_ntt);

/* npoly fixed polynomials (canonical coefficients, npoly x n) prepared once:
 *   RingLPR pk=(a,b), sk=s                    ring_lpr.rs:349-351
 *   PSFGPVRing parity-check row a (k+2 polys) primitive/psf/gpv_ring.rs:66-92 */
int qf_rkey_upload_host(qf_ctx *ctx, const qf_ring *ring, const uint32_t *polys_host,
                        uint32_t npoly, qf_rkey **out);
int qf_rkey_upload_dev(qf_ctx *ctx, const qf_ring *ring, const uint32_t *polys_dev,
                       uint32_t npoly, qf_rkey **out);
void qf_rkey_destroy(qf_rkey *rkey);

/* Kernel (b): B independent products c_i = a_i * b_i in R_q.
 * &PolynomialRingZq * &PolynomialRingZq   ring_lpr.rs:391,455,457,491 ; gpv_ring.rs:238 */
int qf_polymul_dev(qf_ctx *ctx, const qf_ring *ring, const uint32_t *a_dev, const uint32_t *b_dev,
                   uint64_t B, uint32_t *c_dev);

/* RingLPR::enc for B messages (ring_lpr.rs:422-461), pk = rkey polys {0:a, 1:b}:
 *   u_i = a*r_i + e1_i ; v_i = b*r_i + e2_i + mu_i(X)*floor(q/2)
 * mu_bits_dev: B rows of ceil(n/8) bytes, bit k of row i = coefficient k of mu_i, i.e. the
 * bits of |message| mod 2^n (ring_lpr.rs:424-427,318-329).  mu_bits_dev must be 4-byte aligned
 * (here and in qf_ringlpr_dec_dev); all polynomial operands are arrays of uint32_t. */
int qf_ringlpr_enc_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *pk, const uint32_t *r_dev,
                       const uint32_t *e1_dev, const uint32_t *e2_dev, const uint8_t *mu_bits_dev,
                       uint64_t B, uint32_t *u_dev, uint32_t *v_dev);
/* RingLPR::dec (ring_lpr.rs:489-508), sk = rkey poly 0:
 *   w = v - s*u ; bit_k = [dist(w_k, floor(q/2)) < dist(w_k, 0)] ; output packed like mu_bits. */
int qf_ringlpr_dec_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *sk, const uint32_t *u_dev,
                       const uint32_t *v_dev, uint64_t B, uint8_t *mu_bits_dev);

/* GPV-ring FDH verification core for B signatures (signature/fdh.rs:115-123 minus the string
 * hash; primitive/psf/gpv_ring.rs:235-239 f_a and :266-275 check_domain):
 *   t_b = sum_j a_j * (sigma_b[j] mod q) ;  ok_b = [ sum sigma_b^2 <= norm_bound ] && [ t_b == u_b ]
 * a = rkey with mbar = k+2 polys; sigma_dev: B x mbar x n signed coefficients (MatPolyOverZ);
 * u_dev: B x n hashed targets (hash/sha256.rs:249-256); norm_bound = floor(s^2 * n*mbar).
 * t_dev (B x n) may be NULL. */
int qf_fdh_ring_verify_dev(qf_ctx *ctx, const qf_ring *ring, const qf_rkey *a, const int32_t *sigma_dev,
                           const uint32_t *u_dev, uint64_t norm_bound, uint64_t B, uint32_t *t_dev,
                           uint8_t *ok_dev);

/* Hash-to-Z_q^{rows x cols} for B strings (the caller step of FDH::vfy, signature/fdh.rs:119):
 *   out[b][i*cols + j] = hash_to_mat_zq_sha256(msg_b, rows, cols, q)[i][j]     hash/sha256.rs:110-133
 *                      = int(sha256_hex("0 {q} {i} {j} {rows} {cols} {msg_b}"), 16) mod q   (:65-83)
 * With rows = n, cols = 1 this is HashMatPolynomialRingZq{rows:1, cols:1}.hash (hash/sha256.rs:249-256,
 * fdh/gpv_ring.rs:78-82): out[b][r] = coefficient r of the hashed ring element, i.e. the u_dev
 * operand of qf_fdh_ring_verify_dev.  msgs_dev: the UTF-8 bytes of all strings back to back;
 * offsets_dev: B+1 byte offsets (string b = msgs[offsets[b] .. offsets[b+1])). */
int qf_hash_to_mat_zq_sha256_dev(qf_ctx *ctx, const uint8_t *msgs_dev, const uint64_t *offsets_dev, uint64_t B,
                                 uint32_t rows, uint32_t cols, uint32_t q, uint32_t *out_dev);

/* ------------------------------------------------------------------ seeded generator
 * The reference samples from an unseeded RNG (e.g. regev.rs:352-364,401), so parity is defined
 * on inputs produced by this counter-mode SplitMix64 generator, implemented identically in
 * oracle/qf_oracle.c:  value = splitmix64((seed ^ (stream << 48)) + idx).
 *   uniform : (value * q) >> 64                       for idx = first .. first+count-1
 *   bits    : row r takes W64 = ceil(cols/64) values; bit j = bit (j & 63) of value at idx
 *             (first_row+r)*W64 + (j >> 6); packed LSB-first, padding bits 0
 *   noise   : sum of 4 uniform integers in [-W,W] (idx*4+t), stored mod q  (Gaussian stand-in) */
int qf_sample_uniform_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count,
                          uint32_t q, uint32_t *out_dev);
int qf_sample_bits_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first_row, uint64_t rows,
                       uint32_t cols, uint32_t stride_bytes, uint8_t *out_dev);
/* the same bit rows written k-tile-major (QF_X_BITS_KT planes, kt_stride_bytes >= 16 rows apart): x for Regev::enc
 * (regev.rs:401) in the form kernel (a) reads fastest, without a layout pass in between */
int qf_sample_bits_kt_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first_row, uint64_t rows,
                          uint32_t cols, size_t kt_stride_bytes, uint8_t *out_dev);
int qf_sample_noise_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count,
                        uint32_t q, uint32_t W, uint32_t *out_dev);

/* Discrete Gaussian over Z with center 0 and Gaussian parameter s, stored mod q -- the distribution
 * of MatZq::sample_discrete_gauss(rows, cols, q, n, 0, s) at the reference's call sites
 * (pk_encryption/regev.rs:356-363, lpr.rs:342-360,399-417, ring_lpr.rs:430-452): rho_s(x) =
 * exp(-pi x^2 / s^2) on the support |x| <= ceil(s*log2(n)) of GPV08 SampleZ.  Sampled by inversion:
 * sample idx is the x whose cumulative-probability threshold floor(2^64 * P[X <= x]) is the first one
 * above the generator's 64-bit value(stream, first + i).  The table (long double sums from -T up) is
 * built on the host and cached in the context.  Parity with the reference is distributional only
 * (its RNG is unseeded); oracle/qf_oracle.c restates the same table and draw. */
int qf_sample_discrete_gauss_dev(qf_ctx *ctx, uint64_t seed, uint32_t stream, uint64_t first, uint64_t count,
                                 uint32_t q, uint32_t n, double s, uint32_t *out_dev);

/* The same four samplers under a 256-bit KEY instead of a 64-bit seed -- what a host object constructed
 * WITHOUT a seed draws from.  The reference takes every sample from the OS CSPRNG (regev.rs:352-364,401);
 * SplitMix64 above is a test/bench generator whose 64-bit state can be recovered from a published
 * matrix, so it must never produce a secret next to a public value.  Here
 *   value(stream, idx) = first 8 bytes, little-endian, of the ChaCha20 block (20 rounds, D. J. Bernstein's
 *   64-bit-counter form) with key key32, block counter idx (state words 12-13) and nonce
 *   nonce ^ (stream << 48) (state words 14-15),
 * and everything after `value` (multiply-high, bit rows, Irwin-Hall sum, inversion table) is as above.
 * key32: 32 bytes from the host's CSPRNG (getrandom / os.urandom / std::random_device), read during the
 * call only.  oracle/qf_oracle.c restates it (pinned by RFC 7539's block test vector). */
int qf_sample_uniform_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first,
                                uint64_t count, uint32_t q, uint32_t *out_dev);
int qf_sample_bits_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first_row,
                             uint64_t rows, uint32_t cols, uint32_t stride_bytes, uint8_t *out_dev);
int qf_sample_bits_kt_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first_row,
                                uint64_t rows, uint32_t cols, size_t kt_stride_bytes, uint8_t *out_dev);
int qf_sample_noise_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first,
                              uint64_t count, uint32_t q, uint32_t W, uint32_t *out_dev);
int qf_sample_discrete_gauss_keyed_dev(qf_ctx *ctx, const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first,
                                       uint64_t count, uint32_t q, uint32_t n, double s, uint32_t *out_dev);

/* ------------------------------------------------------------------ introspection for benches
 * Average duration (ms) of the LAST launch of the dominant kernel of the named op, measured
 * with HIP events on the context's stream when timing is enabled. */
int qf_set_timing(qf_ctx *ctx, int enabled);
int qf_last_kernel_ms(qf_ctx *ctx, const char *which /* "gemm","ntt","matvec","decode" */, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* QFGPU_H */
