// qfall_crypto.hpp -- C++17 host side of the MI355X hot path, mirroring the reference's trait API
// (paths relative to /root/reference/src).  The reference is Rust; this image has no Rust
// toolchain, so the host layer that is built and tested above the C ABI (include/qfgpu.h) is
// this header: same names, argument meaning and error behaviour as
//   PKEncryptionScheme / GenericMultiBitEncryption   construction/pk_encryption.rs:55-167
//   Regev, DualRegev, LPR, RingLPR                   construction/pk_encryption/{regev,dual_regev,lpr,ring_lpr}.rs
//   SISHash                                          construction/hash/sis.rs:52-167
//   FDH<PSFGPVRing>::vfy, PSFGPVRing::f_a/check_domain   construction/signature/fdh.rs:115-123,
//                                                    primitive/psf/gpv_ring.rs:235-275
// Where the reference panics, these throw qfall::Panic; where it returns Err(MathError), they
// throw qfall::MathError.  Header-only; link against crypto_amd/lib/libqfgpu.so.  Nothing here
// computes lattice arithmetic on the CPU: every product goes through the C ABI.
//
// Randomness: the reference samples from the OS's CSPRNG (regev.rs:352-364,401).  Here every scheme
// object owns a counter-mode stream set (qfgpu.h "seeded generator"): under an explicit 64-bit seed
// SplitMix64 (reproducible, comparable with the oracle, NOT confidential), and -- the default
// construction -- ChaCha20 under a fresh 256-bit key from the OS; explicit randomness can be passed to enc.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "qfgpu.h"

namespace qfall {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
struct MathError : std::runtime_error { using std::runtime_error::runtime_error; };

constexpr uint64_t MASTER_SEED = 0x5146414C4CULL;  // "QFALL": what tests and benches pass EXPLICITLY; never a default
// What a scheme's generator runs under.  An explicit 64-bit seed selects SplitMix64 in counter mode: a
// benchmark/test generator whose state can be recovered from a published matrix -- passing one is a
// statement that confidentiality is not wanted.  Seed::fresh(), the default of every constructor,
// selects ChaCha20 under 256 bits from the OS (std::random_device reads the OS's CSPRNG on Linux), as
// the reference draws OS randomness at every call (regev.rs:352-364,401): the public matrix then
// reveals nothing about the secret, noise or encryption streams of the same object.
struct Seed {
    bool keyed = false;
    uint64_t seed = 0;       // SplitMix64 seed, or the ChaCha20 nonce
    uint8_t key[32] = {0};   // ChaCha20 key (keyed only)
    Seed(uint64_t s) : seed(s) {}   // implicit on purpose: Regev(dev, n, m, q, alpha, MASTER_SEED)
    static Seed fresh() {
        Seed k(0);
        k.keyed = true;
        std::random_device rd;
        for (int i = 0; i < 32; i += 4) {
            const uint32_t w = rd();
            std::memcpy(k.key + i, &w, 4);
        }
        return k;
    }
};

// ---------------------------------------------------------------------------------------------
// value types at the seam (qfall-math MatZq / MatZ / PolynomialRingZq / MatPolyOverZ)
// ---------------------------------------------------------------------------------------------
struct MatZq {
    uint32_t rows = 0, cols = 0, q = 0;
    std::vector<uint32_t> e;  // row-major canonical residues
    MatZq() = default;
    MatZq(uint32_t r, uint32_t c, uint32_t q_) : rows(r), cols(c), q(q_), e((size_t)r * c, 0) {
        if (q_ <= 1) throw Panic("modulus must be > 1");
    }
    uint32_t get_num_rows() const { return rows; }
    uint32_t get_num_columns() const { return cols; }
    uint32_t get_mod() const { return q; }
    bool is_column_vector() const { return cols == 1; }
    // negative indices count from the end, as get_entry(-1, 0) in regev.rs:410
    uint32_t &at(long i, long j) {
        if (i < 0) i += rows;
        if (j < 0) j += cols;
        if (i < 0 || j < 0 || i >= (long)rows || j >= (long)cols) throw Panic("matrix index out of bounds");
        return e[(size_t)i * cols + j];
    }
    uint32_t get_entry(long i, long j) const { return const_cast<MatZq *>(this)->at(i, j); }
    void set_entry(long i, long j, uint64_t v) { at(i, j) = (uint32_t)(v % q); }
    bool operator==(const MatZq &o) const { return rows == o.rows && cols == o.cols && q == o.q && e == o.e; }
    std::string to_string() const {  // "[[a, b],[c, d]] mod q"  (sha256.rs:328, sis.rs:152)
        std::string s = "[";
        for (uint32_t i = 0; i < rows; i++) {
            s += i ? ",[" : "[";
            for (uint32_t j = 0; j < cols; j++) s += (j ? ", " : "") + std::to_string(e[(size_t)i * cols + j]);
            s += "]";
        }
        return s + "] mod " + std::to_string(q);
    }
};
struct MatZ {  // small signed integers (DualRegev secret key x in {0,1}^m)
    uint32_t rows = 0, cols = 0;
    std::vector<int64_t> e;
};
struct PolynomialRingZq {  // element of Z_q[X]/(X^n+1), degree-ascending canonical coefficients
    uint32_t n = 0, q = 0;
    std::vector<uint32_t> c;
    bool operator==(const PolynomialRingZq &o) const { return n == o.n && q == o.q && c == o.c; }
};
struct MatPolyOverZ {  // column vector of signed integer polynomials (FDH-ring signature)
    uint32_t rows = 0, n = 0;
    std::vector<int32_t> c;  // rows x n
};

// ---------------------------------------------------------------------------------------------
// device plumbing
// ---------------------------------------------------------------------------------------------
class Device {
  public:
    explicit Device(int device = 0) {
        if (qf_ctx_create(device, nullptr, &ctx_) != QF_OK) throw Panic("no HIP device: there is no CPU fallback");
    }
    ~Device() {
        for (auto &c : cache_) qf_key_destroy(c.k);
        qf_ctx_destroy(ctx_);
    }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    qf_ctx *ctx() const { return ctx_; }
    // The digit planes of a public matrix are derived data: upload + recode once per DISTINCT matrix and
    // keep the handle (the reference passes `&pk` to every enc; re-uploading 6 MB per call would dominate
    // single-message calls).  Hits are confirmed by comparing the full contents, so a stale or colliding
    // entry can never be used; the 8 most recently used matrices stay resident.
    qf_key *cached_key(const MatZq &m, bool transpose = false) const {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : m.e) { h ^= v; h *= 1099511628211ull; }
        for (auto &c : cache_)
            if (c.hash == h && c.rows == m.rows && c.cols == m.cols && c.q == m.q && c.tr == transpose &&
                c.e.size() == m.e.size() && std::memcmp(c.e.data(), m.e.data(), m.e.size() * 4) == 0) {
                c.stamp = ++clock_;
                return c.k;
            }
        qf_key *k = nullptr;
        check(qf_key_upload_host(ctx_, m.e.data(), m.rows, m.cols, m.q, transpose ? 1 : 0, &k));
        if (cache_.size() >= 8) {
            size_t old = 0;
            for (size_t i = 1; i < cache_.size(); i++) if (cache_[i].stamp < cache_[old].stamp) old = i;
            qf_key_destroy(cache_[old].k);
            cache_.erase(cache_.begin() + (long)old);
        }
        cache_.push_back(KeyEntry{m.rows, m.cols, m.q, transpose, h, m.e, k, ++clock_});
        return k;
    }
    size_t cached_keys() const { return cache_.size(); }
    void check(int rc) const {
        if (rc != QF_OK) throw Panic(std::string("qfgpu error ") + std::to_string(rc) + ": " + qf_last_error(ctx_));
    }
    void sync() const { check(qf_sync(ctx_)); }

  private:
    struct KeyEntry { uint32_t rows, cols, q; bool tr; uint64_t hash; std::vector<uint32_t> e; qf_key *k; uint64_t stamp; };
    qf_ctx *ctx_ = nullptr;
    mutable std::vector<KeyEntry> cache_;
    mutable uint64_t clock_ = 0;
};

template <class T>
class DevBuf {  // owning device array
  public:
    DevBuf(const Device &d, size_t count) : dev_(&d), n_(count) {
        if (count) d.check(qf_malloc(d.ctx(), count * sizeof(T), (void **)&p_));
    }
    DevBuf(const Device &d, const std::vector<T> &h) : DevBuf(d, h.size()) { upload(h.data(), h.size()); }
    ~DevBuf() { if (p_) qf_free(dev_->ctx(), p_); }
    DevBuf(DevBuf &&o) noexcept : dev_(o.dev_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DevBuf(const DevBuf &) = delete;
    T *get() const { return p_; }
    size_t size() const { return n_; }
    void upload(const T *h, size_t count) { if (count) dev_->check(qf_memcpy_h2d(dev_->ctx(), p_, h, count * sizeof(T))); }
    std::vector<T> download() const {
        std::vector<T> h(n_);
        if (n_) dev_->check(qf_memcpy_d2h(dev_->ctx(), h.data(), p_, n_ * sizeof(T)));
        return h;
    }

  private:
    const Device *dev_;
    T *p_ = nullptr;
    size_t n_ = 0;
};

class DevKey {  // MatZq resident on the device (qf_key)
  public:
    DevKey(const Device &d, const MatZq &m, bool transpose = false) : dev_(&d) {
        d.check(qf_key_upload_host(d.ctx(), m.e.data(), m.rows, m.cols, m.q, transpose ? 1 : 0, &k_));
    }
    ~DevKey() { qf_key_destroy(k_); }
    DevKey(const DevKey &) = delete;
    qf_key *get() const { return k_; }

  private:
    const Device *dev_;
    qf_key *k_ = nullptr;
};

// seeded draws (streams: 1 uniform A/a, 2 secret, 3 x/r, 4 noise e, 5 messages, 6 free operands)
class Randomness {
  public:
    Randomness(const Device &d, Seed seed = Seed::fresh()) : dev_(&d), seed_(seed) {}
    const Seed &seed() const { return seed_; }
    uint64_t take(uint32_t stream, uint64_t count) { uint64_t f = pos_[stream & 15]; pos_[stream & 15] += count; return f; }
    // ---- draws left in device memory (key generation keeps A, s and e there)
    void uniform_dev(uint32_t stream, uint64_t count, uint32_t q, uint32_t *out_dev) {
        const uint64_t f = take(stream, count);
        dev_->check(seed_.keyed ? qf_sample_uniform_keyed_dev(dev_->ctx(), seed_.key, seed_.seed, stream, f, count, q, out_dev)
                                : qf_sample_uniform_dev(dev_->ctx(), seed_.seed, stream, f, count, q, out_dev));
    }
    void gauss_dev(uint32_t stream, uint64_t count, uint32_t q, uint32_t n, double s, uint32_t *out_dev) {
        const uint64_t f = take(stream, count);
        dev_->check(seed_.keyed ? qf_sample_discrete_gauss_keyed_dev(dev_->ctx(), seed_.key, seed_.seed, stream, f, count, q, n, s, out_dev)
                                : qf_sample_discrete_gauss_dev(dev_->ctx(), seed_.seed, stream, f, count, q, n, s, out_dev));
    }
    // chi: D_{Z,s} as MatZq::sample_discrete_gauss(.., q, n, 0, s) draws it (regev.rs:356-363,
    // lpr.rs:342-360) -- the schemes select it with (n, alpha*q) when they are constructed -- or, after
    // use_irwin_hall(), the benchmark generator's bounded stand-in (sum of 4 uniforms in [-W, W])
    void use_discrete_gauss(uint32_t n, double s) { gauss_n_ = n; gauss_s_ = s; }
    void use_irwin_hall() { gauss_n_ = 0; }
    void noise_dev(uint32_t stream, uint64_t count, uint32_t q, uint32_t W, uint32_t *out_dev) {
        if (gauss_n_) return gauss_dev(stream, count, q, gauss_n_, gauss_s_, out_dev);
        const uint64_t f = take(stream, count);
        dev_->check(seed_.keyed ? qf_sample_noise_keyed_dev(dev_->ctx(), seed_.key, seed_.seed, stream, f, count, q, W, out_dev)
                                : qf_sample_noise_dev(dev_->ctx(), seed_.seed, stream, f, count, q, W, out_dev));
    }
    void bits_dev(uint32_t stream, uint64_t rows, uint32_t cols, uint32_t stride, uint8_t *out_dev) {
        const uint64_t f = take(stream, rows);
        dev_->check(seed_.keyed ? qf_sample_bits_keyed_dev(dev_->ctx(), seed_.key, seed_.seed, stream, f, rows, cols, stride, out_dev)
                                : qf_sample_bits_dev(dev_->ctx(), seed_.seed, stream, f, rows, cols, stride, out_dev));
    }
    // the same rows written k-tile-major (QF_X_BITS_KT planes, kt_stride bytes apart)
    void bits_kt_dev(uint32_t stream, uint64_t rows, uint32_t cols, size_t kt_stride, uint8_t *out_dev) {
        const uint64_t f = take(stream, rows);
        dev_->check(seed_.keyed ? qf_sample_bits_kt_keyed_dev(dev_->ctx(), seed_.key, seed_.seed, stream, f, rows, cols, kt_stride, out_dev)
                                : qf_sample_bits_kt_dev(dev_->ctx(), seed_.seed, stream, f, rows, cols, kt_stride, out_dev));
    }
    // ---- the same draws brought to the host
    std::vector<uint32_t> uniform(uint32_t stream, uint64_t count, uint32_t q) {
        DevBuf<uint32_t> b(*dev_, count);
        uniform_dev(stream, count, q, b.get());
        return b.download();
    }
    std::vector<uint32_t> gauss(uint32_t stream, uint64_t count, uint32_t q, uint32_t n, double s) {
        DevBuf<uint32_t> b(*dev_, count);
        gauss_dev(stream, count, q, n, s, b.get());
        return b.download();
    }
    std::vector<uint32_t> noise(uint32_t stream, uint64_t count, uint32_t q, uint32_t W) {
        DevBuf<uint32_t> b(*dev_, count);
        noise_dev(stream, count, q, W, b.get());
        return b.download();
    }
    std::vector<uint32_t> bits(uint32_t stream, uint32_t count) {  // MatZ::sample_uniform(m,1,0,2)
        const uint32_t stride = ((count + 127) / 128) * 16;
        DevBuf<uint8_t> b(*dev_, stride);
        bits_dev(stream, 1, count, stride, b.get());
        std::vector<uint8_t> p = b.download();
        std::vector<uint32_t> out(count);
        for (uint32_t i = 0; i < count; i++) out[i] = (p[i >> 3] >> (i & 7)) & 1u;
        return out;
    }

  private:
    const Device *dev_;
    Seed seed_;
    uint64_t pos_[16] = {0};
    uint32_t gauss_n_ = 0;
    double gauss_s_ = 0.0;
};

// chi of a freshly constructed scheme: the reference's D_{Z, alpha*q} whenever the sampler supports the
// parameters (n >= 2, support s*log2(n) in [1, 2^21]); otherwise the bounded stand-in stays
inline void default_noise(Randomness &rng, uint32_t n, double s) {
    if (n >= 2 && s > 0.0) {
        const double support = std::ceil(s * std::log2((double)n));
        if (support >= 1.0 && support <= (double)(1 << 21)) rng.use_discrete_gauss(n, s);
    }
}

// ---------------------------------------------------------------------------------------------
// helpers shared by the parameter rules
// ---------------------------------------------------------------------------------------------
inline bool is_prime(uint64_t n) {
    if (n < 2) return false;
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull}) { if (n % p == 0) return n == p; }
    for (uint64_t d = 17; d * d <= n; d += 2) if (n % d == 0) return false;
    return true;
}
// deterministic stand-in for Z::sample_prime_uniform(lower, upper): the largest prime in range
inline uint64_t largest_prime_at_most(uint64_t upper, uint64_t lower) {
    for (uint64_t p = upper; p >= lower && p >= 2; p--) if (is_prime(p)) return p;
    throw MathError("no prime in range");
}
inline uint32_t log2_ceil(uint64_t q) { uint32_t k = 0; while ((1ull << k) < q) k++; return k; }  // log_ceil(2)
inline uint64_t ipow(uint64_t b, int e) { uint64_t r = 1; while (e-- > 0) r *= b; return r; }
// half-width of the seeded Gaussian stand-in for Gaussian parameter s = alpha*q
inline uint32_t gaussian_width(double s) { return (uint32_t)std::llround(s / std::sqrt(2 * M_PI) * std::sqrt(3.0) / 2); }
// message.modulo(2): Euclidean remainder, also for i64::MIN (regev.rs:398,598-612)
inline uint32_t message_bit(int64_t m) { return (uint32_t)(((m % 2) + 2) % 2); }
inline uint32_t bit_stride(uint32_t cols) { return ((cols + 127) / 128) * 16; }

// ---------------------------------------------------------------------------------------------
// PKEncryptionScheme (pk_encryption.rs:55-82) + GenericMultiBitEncryption (:120-167), CRTP
// ---------------------------------------------------------------------------------------------
template <class Derived, class PK, class SK, class Cipher>
struct PKEncryptionScheme {
    using PublicKey = PK; using SecretKey = SK; using CipherT = Cipher;
    // LSB-first bitwise encryption of |message| (pk_encryption.rs:130-144)
    std::vector<Cipher> enc_multiple_bits(const PK &pk, int64_t message) {
        uint64_t v = message < 0 ? (uint64_t)(-(message + 1)) + 1 : (uint64_t)message;
        std::vector<Cipher> out;
        for (; v; v >>= 1) out.push_back(static_cast<Derived *>(this)->enc(pk, (int64_t)(v & 1)));
        return out;
    }
    uint64_t dec_multiple_bits(const SK &sk, const std::vector<Cipher> &cipher) {  // :154-166
        uint64_t out = 0;
        for (size_t i = 0; i < cipher.size(); i++)
            if (static_cast<Derived *>(this)->dec(sk, cipher[i]) != 0) out |= 1ull << i;
        return out;
    }
};

// batched decryption of the three matrix schemes: bits = rule(<w, c[:-1]> + c[-1])
inline std::vector<uint8_t> decode_batch(const Device &d, const std::vector<uint32_t> &w, uint32_t q,
                                         const uint32_t *C_dev, uint64_t B) {
    DevBuf<uint32_t> dw(d, w);
    DevBuf<uint8_t> bits(d, B);
    d.check(qf_dot_decode_dev(d.ctx(), dw.get(), (uint32_t)w.size(), q, C_dev, B, bits.get()));
    return bits.download();
}

// pk = [A ; s^t A + e^t], sk = s (regev.rs:350-374 with s uniform; lpr.rs:338-370 with s <- chi): A, s and e
// are drawn into ONE device image [pk | s], the product runs on it (qf_vecmat_u32_dev), one copy comes back
inline std::pair<MatZq, MatZq> lwe_keypair(const Device &d, Randomness &rng, uint32_t n, uint32_t m, uint32_t q,
                                           bool secret_from_chi, uint32_t W) {
    const size_t pkw = (size_t)(n + 1) * m;
    DevBuf<uint32_t> img(d, pkw + n), e(d, m);
    uint32_t *A = img.get(), *s = img.get() + pkw;
    rng.uniform_dev(1, (uint64_t)n * m, q, A);
    if (secret_from_chi) rng.noise_dev(2, n, q, W, s); else rng.uniform_dev(2, n, q, s);
    rng.noise_dev(4, m, q, W, e.get());
    d.check(qf_vecmat_u32_dev(d.ctx(), A, n, m, q, s, e.get(), A + (size_t)n * m));
    const std::vector<uint32_t> h = img.download();
    MatZq pk(n + 1, m, q), sk(n, 1, q);
    std::copy(h.begin(), h.begin() + pkw, pk.e.begin());
    std::copy(h.begin() + pkw, h.end(), sk.e.begin());
    return {pk, sk};
}

// ---------------------------------------------------------------------------------------------
// Regev (regev.rs)
// ---------------------------------------------------------------------------------------------
class Regev : public PKEncryptionScheme<Regev, MatZq, MatZq, MatZq> {
  public:
    uint32_t n, m, q;
    double alpha;
    Regev(const Device &d, uint32_t n_, uint32_t m_, uint32_t q_, double alpha_, Seed seed = Seed::fresh())
        : n(n_), m(m_), q(q_), alpha(alpha_), dev(&d), rng(d, seed) {            // Regev::new regev.rs:87-99
        if (q <= 1) throw Panic("modulus must be > 1");
        default_noise(rng, n, alpha * q);
    }
    static Regev default_(const Device &d) { return Regev(d, 13, 154, 1427, 0.01); }          // regev.rs:317-324
    static Regev secure128(const Device &d) { return Regev(d, 230, 5313, 7764299, 0.0011); }  // regev.rs:301-303
    static Regev new_from_n(const Device &d, uint32_t n) {                                    // regev.rs:122-149
        if (n < 10) throw Panic("Choose n >= 10 as this function does not return parameters ensuring proper correctness of the scheme otherwise.");
        const int power = (n >= 2 && n <= 4) ? 5 : n == 5 ? 4 : 3;                            // regev.rs:174-201
        const uint64_t ub = ipow(n, power), lb = (ub + 1) / 2;
        uint64_t upper = ub;
        for (;;) {
            const uint64_t qq = largest_prime_at_most(upper, lb);
            Regev out(d, n, (n + 1) * log2_ceil(qq), (uint32_t)qq, 1.0 / (2 * std::sqrt((double)n) * std::pow(std::log2((double)n), 2)));
            try { out.check_correctness(); out.check_security(); return out; } catch (const MathError &) { upper = qq - 1; }
        }
    }
    void check_correctness() const {  // regev.rs:229-253
        if (n <= 1) throw MathError("n must be chosen bigger than 1.");
        if (alpha > 1 / (std::sqrt((double)n) * std::log2((double)n))) throw MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log n)");
        if (20 * std::sqrt((double)m) * alpha > q) throw MathError("Correctness is not guaranteed as 5 * sqrt(m) * alpha > q/4");
    }
    void check_security() const {  // regev.rs:277-295
        if (q * alpha < 2 * std::sqrt((double)n)) throw MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)");
        if (m <= std::ceil((n + 1) * std::log2((double)q))) throw MathError("Security is not guaranteed as m <= (n + 1) log q");
    }
    uint32_t noise_width() const { return gaussian_width(alpha * q); }

    // regev.rs:350-374: A <- Z_q^{n x m}, s <- Z_q^n, e <- chi^m, b^t = s^t A + e^t, pk = [A ; b^t]
    std::pair<MatZq, MatZq> gen() { return lwe_keypair(*dev, rng, n, m, q, false, noise_width()); }
    // regev.rs:396-414: c = pk*x + [0..0, (message mod 2)*floor(q/2)]^t, x <- {0,1}^m
    MatZq enc(const MatZq &pk, int64_t message, const std::vector<uint32_t> *x_in = nullptr) {
        check_pk(pk);
        std::vector<uint32_t> x = x_in ? *x_in : rng.bits(3, m);
        if (x.size() != m) throw Panic("mismatching matrix dimensions");
        MatZq c(n + 1, 1, q);
        qf_key *k = dev->cached_key(pk);
        dev->check(qf_matvec_modq_host(dev->ctx(), k, x.data(), c.e.data()));
        c.set_entry(-1, 0, (uint64_t)c.get_entry(-1, 0) + (uint64_t)message_bit(message) * (q / 2));
        return c;
    }
    // regev.rs:438-452
    int64_t dec(const MatZq &sk, const MatZq &cipher) {
        if (cipher.rows != n + 1 || cipher.cols != 1) throw Panic("mismatching matrix dimensions");
        DevBuf<uint32_t> dc(*dev, cipher.e);
        return dec_batch(sk, dc.get(), 1)[0];
    }
    // B encryptions on kernel (a).  x_bits: B rows of bit_stride(m) bytes (LSB-first), on the host.
    // Returns B x (n+1) ciphertexts on the device.
    DevBuf<uint32_t> enc_batch(const MatZq &pk, const std::vector<uint8_t> &msg_bits, const std::vector<uint8_t> &x_bits) {
        check_pk(pk);
        const uint64_t B = msg_bits.size();
        if (x_bits.size() != B * bit_stride(m)) throw Panic("mismatching randomness dimensions");
        qf_key *k = dev->cached_key(pk);
        DevBuf<uint8_t> dx(*dev, x_bits), dm(*dev, msg_bits);
        DevBuf<uint32_t> out(*dev, B * (n + 1));
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, dx.get(), QF_X_BITS, bit_stride(m), 1, B, nullptr, dm.get(), out.get()));
        return out;
    }
    // The same with x drawn on the device as Regev::enc does (regev.rs:401): stream 3 of the scheme's generator, written
    // k-tile-major (the form kernel (a) reads fastest: no layout pass, 3.5-4 % faster product).
    DevBuf<uint32_t> enc_batch(const MatZq &pk, const std::vector<uint8_t> &msg_bits) {
        check_pk(pk);
        const uint64_t B = msg_bits.size();
        qf_key *k = dev->cached_key(pk);
        const size_t planes = ((size_t)m + 127) / 128, kt_stride = (size_t)B * 16;
        DevBuf<uint8_t> dx(*dev, planes * kt_stride), dm(*dev, msg_bits);
        DevBuf<uint32_t> out(*dev, B * (n + 1));
        rng.bits_kt_dev(3, B, m, kt_stride, dx.get());
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, dx.get(), QF_X_BITS_KT, kt_stride, 1, B, nullptr, dm.get(), out.get()));
        return out;
    }
    std::vector<uint8_t> dec_batch(const MatZq &sk, const uint32_t *C_dev, uint64_t B) {
        if (sk.rows != n || sk.cols != 1) throw Panic("mismatching secret key dimension");
        std::vector<uint32_t> w(n);
        for (uint32_t i = 0; i < n; i++) w[i] = (q - sk.e[i] % q) % q;  // Z::MINUS_ONE * sk
        return decode_batch(*dev, w, q, C_dev, B);
    }
    Randomness &randomness() { return rng; }

  protected:
    void check_pk(const MatZq &pk) const {
        if (pk.q != q) throw Panic("mismatching modulus between scheme and key");
        if (pk.rows != n + 1 || pk.cols != m) throw Panic("mismatching matrix dimensions");
    }
    const Device *dev;
    Randomness rng;
};

// ---------------------------------------------------------------------------------------------
// DualRegev (dual_regev.rs): pk = [A | u], u = A*x, sk = x in {0,1}^m
// ---------------------------------------------------------------------------------------------
class DualRegev : public PKEncryptionScheme<DualRegev, MatZq, MatZ, MatZq> {
  public:
    uint32_t n, m, q;
    double alpha;
    DualRegev(const Device &d, uint32_t n_, uint32_t m_, uint32_t q_, double alpha_, Seed seed = Seed::fresh())
        : n(n_), m(m_), q(q_), alpha(alpha_), dev(&d), rng(d, seed) { if (q <= 1) throw Panic("modulus must be > 1"); default_noise(rng, n, alpha * q); }
    static DualRegev default_(const Device &d) { return DualRegev(d, 13, 154, 1427, 0.01); }
    static DualRegev new_from_n(const Device &d, uint32_t n) {   // dual_regev.rs:122-201 repeats regev.rs
        Regev r = Regev::new_from_n(d, n);
        return DualRegev(d, r.n, r.m, r.q, r.alpha);
    }
    uint32_t noise_width() const { return gaussian_width(alpha * q); }
    std::pair<MatZq, MatZ> gen() {  // dual_regev.rs:347-361
        MatZq A(n, m, q);
        A.e = rng.uniform(1, (uint64_t)n * m, q);
        std::vector<uint32_t> x = rng.bits(2, m), u(n);
        { DevKey ka(*dev, A); dev->check(qf_matvec_modq_host(dev->ctx(), ka.get(), x.data(), u.data())); }
        MatZq pk(n, m + 1, q);
        for (uint32_t i = 0; i < n; i++) {
            std::copy(A.e.begin() + (size_t)i * m, A.e.begin() + (size_t)(i + 1) * m, pk.e.begin() + (size_t)i * (m + 1));
            pk.e[(size_t)i * (m + 1) + m] = u[i];
        }
        MatZ sk; sk.rows = m; sk.cols = 1; sk.e.assign(x.begin(), x.end());
        return {pk, sk};
    }
    // dual_regev.rs:385-413: c^t = s^t*pk + e^t, message into the last entry
    MatZq enc(const MatZq &pk, int64_t message, const std::vector<uint32_t> *s_in = nullptr, const std::vector<uint32_t> *e_in = nullptr) {
        if (pk.q != q) throw Panic("mismatching modulus between scheme and key");
        if (pk.rows != n || pk.cols != m + 1) throw Panic("mismatching matrix dimensions");
        std::vector<uint32_t> s = s_in ? *s_in : rng.uniform(3, n, q);
        std::vector<uint32_t> e = e_in ? *e_in : rng.noise(4, m + 1, q, noise_width());
        MatZq c(m + 1, 1, q);
        qf_key *k = dev->cached_key(pk);
        dev->check(qf_vecmat_modq_host(dev->ctx(), k, s.data(), c.e.data()));
        for (uint32_t j = 0; j <= m; j++) c.e[j] = (uint32_t)(((uint64_t)c.e[j] + e[j]) % q);
        c.set_entry(-1, 0, (uint64_t)c.get_entry(-1, 0) + (uint64_t)message_bit(message) * (q / 2));
        return c;
    }
    int64_t dec(const MatZ &sk, const MatZq &cipher) {  // dual_regev.rs:437-450
        if (sk.e.size() != m || cipher.rows != m + 1) throw Panic("mismatching matrix dimensions");
        std::vector<uint32_t> w(m);
        for (uint32_t i = 0; i < m; i++) w[i] = (uint32_t)((((-sk.e[i]) % (int64_t)q) + q) % q);
        DevBuf<uint32_t> dc(*dev, cipher.e);
        return decode_batch(*dev, w, q, dc.get(), 1)[0];
    }
    // batched: C = S*pk + E with the key uploaded transposed (kernel a at full width)
    DevBuf<uint32_t> enc_batch(const MatZq &pk, const std::vector<uint8_t> &msg_bits, const std::vector<uint32_t> &S, const std::vector<uint32_t> &E) {
        const uint64_t B = msg_bits.size();
        if (S.size() != B * n || E.size() != B * (m + 1)) throw Panic("mismatching randomness dimensions");
        qf_key *k = dev->cached_key(pk, true);
        DevBuf<uint32_t> ds(*dev, S), de(*dev, E), out(*dev, B * (m + 1));
        DevBuf<uint8_t> dm(*dev, msg_bits);
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, ds.get(), QF_X_U32, (size_t)n * 4, 0, B, de.get(), dm.get(), out.get()));
        return out;
    }
    Randomness &randomness() { return rng; }

  private:
    const Device *dev;
    Randomness rng;
};

// ---------------------------------------------------------------------------------------------
// LPR (lpr.rs): pk = [A ; b^t] (n+1) x n, c = pk*r + e
// ---------------------------------------------------------------------------------------------
inline int lpr_factor(uint32_t n) { return n <= 20 ? 1 : n <= 40 ? 2 : n <= 80 ? 3 : n <= 160 ? 4 : 5; }  // lpr.rs:178-184

class LPR : public PKEncryptionScheme<LPR, MatZq, MatZq, MatZq> {
  public:
    uint32_t n, q;
    double alpha;
    LPR(const Device &d, uint32_t n_, uint32_t q_, double alpha_, Seed seed = Seed::fresh())
        : n(n_), q(q_), alpha(alpha_), dev(&d), rng(d, seed) { if (q <= 1) throw Panic("modulus must be > 1"); default_noise(rng, n, alpha * q); }
    static LPR default_(const Device &d) { return LPR(d, 10, 983, 0.0072); }            // lpr.rs:306-312
    static LPR secure128(const Device &d) { return LPR(d, 500, 76859609, 0.000005); }   // lpr.rs:290-292
    static LPR new_from_n(const Device &d, uint32_t n) {                                // lpr.rs:113-191
        if (n < 10) throw Panic("Choose n >= 10 as this function does not return parameters ensuring proper correctness of the scheme otherwise.");
        const uint64_t ub = ipow(n, 3), lb = (ub + 1) / 2;
        uint64_t upper = ub;
        for (;;) {
            const uint64_t qq = largest_prime_at_most(upper, lb);
            LPR out(d, n, (uint32_t)qq, 1.0 / (lpr_factor(n) * std::sqrt((double)n) * std::pow(std::log2((double)n), 3)));
            try { out.check_correctness(); out.check_security(); return out; } catch (const MathError &) { upper = qq - 1; }
        }
    }
    void check_correctness() const {  // lpr.rs:221-248
        if (n <= 1) throw MathError("n must be chosen bigger than 1.");
        if (alpha > 1.0 / (lpr_factor(n) * std::sqrt((double)n) * std::pow(std::log2((double)n), 3)))
            throw MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log^3 n)");
    }
    void check_security() const {  // lpr.rs:272-284
        if (q * alpha < 2 * std::sqrt((double)n)) throw MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)");
    }
    uint32_t noise_width() const { return gaussian_width(alpha * q); }
    int noise_digits() const {  // signed 8-bit digits covering the generator's range |v| <= 4W
        const uint64_t bound = 4ull * noise_width();
        int d = 1;
        uint64_t cap = 127;
        while (cap < bound) { cap = cap * 256 + 127; d++; }
        return d;
    }
    std::pair<MatZq, MatZq> gen() { return lwe_keypair(*dev, rng, n, n, q, true, noise_width()); }  // lpr.rs:338-370
    // lpr.rs:394-430: c = pk*r + e + [0..0, msg*floor(q/2)]^t
    MatZq enc(const MatZq &pk, int64_t message, const std::vector<uint32_t> *r_in = nullptr, const std::vector<uint32_t> *e_in = nullptr) {
        check_pk(pk);
        const uint32_t W = noise_width();
        std::vector<uint32_t> r = r_in ? *r_in : rng.noise(3, n, q, W);
        std::vector<uint32_t> e = e_in ? *e_in : rng.noise(4, n + 1, q, W);
        MatZq c(n + 1, 1, q);
        qf_key *k = dev->cached_key(pk);
        dev->check(qf_matvec_modq_host(dev->ctx(), k, r.data(), c.e.data()));
        for (uint32_t i = 0; i <= n; i++) c.e[i] = (uint32_t)(((uint64_t)c.e[i] + e[i]) % q);
        c.set_entry(-1, 0, (uint64_t)c.get_entry(-1, 0) + (uint64_t)message_bit(message) * (q / 2));
        return c;
    }
    int64_t dec(const MatZq &sk, const MatZq &cipher) {  // lpr.rs:454-468
        if (cipher.rows != n + 1 || cipher.cols != 1) throw Panic("mismatching matrix dimensions");
        DevBuf<uint32_t> dc(*dev, cipher.e);
        return dec_batch(sk, dc.get(), 1)[0];
    }
    DevBuf<uint32_t> enc_batch(const MatZq &pk, const std::vector<uint8_t> &msg_bits, const std::vector<uint32_t> &R, const std::vector<uint32_t> &E, int digits = 0) {
        check_pk(pk);
        const uint64_t B = msg_bits.size();
        if (R.size() != B * n || E.size() != B * (n + 1)) throw Panic("mismatching randomness dimensions");
        qf_key *k = dev->cached_key(pk);
        DevBuf<uint32_t> dr(*dev, R), de(*dev, E), out(*dev, B * (n + 1));
        DevBuf<uint8_t> dm(*dev, msg_bits);
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, dr.get(), QF_X_U32, (size_t)n * 4, digits, B, de.get(), dm.get(), out.get()));
        return out;
    }
    std::vector<uint8_t> dec_batch(const MatZq &sk, const uint32_t *C_dev, uint64_t B) {
        if (sk.rows != n || sk.cols != 1) throw Panic("mismatching secret key dimension");
        std::vector<uint32_t> w(n);
        for (uint32_t i = 0; i < n; i++) w[i] = (q - sk.e[i] % q) % q;
        return decode_batch(*dev, w, q, C_dev, B);
    }
    Randomness &randomness() { return rng; }

  private:
    void check_pk(const MatZq &pk) const {
        if (pk.q != q) throw Panic("mismatching modulus between scheme and key");
        if (pk.rows != n + 1 || pk.cols != n) throw Panic("mismatching matrix dimensions");
    }
    const Device *dev;
    Randomness rng;
};

// ---------------------------------------------------------------------------------------------
// R_q plumbing + RingLPR (ring_lpr.rs)
// ---------------------------------------------------------------------------------------------
class Ring {  // utils/common_moduli.rs:41-48 new_anticyclic: X^n + 1 mod q
  public:
    Ring(const Device &d, uint32_t n, uint32_t q) : dev_(&d), n_(n), q_(q) { d.check(qf_ring_create(d.ctx(), n, q, &r_)); }
    ~Ring() { qf_ring_destroy(r_); }
    Ring(const Ring &) = delete;
    qf_ring *get() const { return r_; }
    uint32_t n() const { return n_; }
    uint32_t q() const { return q_; }
    bool uses_ntt() const { int u = 0; qf_ring_info(r_, nullptr, nullptr, &u); return u != 0; }

  private:
    const Device *dev_;
    qf_ring *r_ = nullptr;
    uint32_t n_, q_;
};
class RingKey {
  public:
    RingKey(const Device &d, const Ring &ring, const std::vector<uint32_t> &polys, uint32_t npoly) {
        d.check(qf_rkey_upload_host(d.ctx(), ring.get(), polys.data(), npoly, &k_));
    }
    ~RingKey() { qf_rkey_destroy(k_); }
    RingKey(const RingKey &) = delete;
    qf_rkey *get() const { return k_; }

  private:
    qf_rkey *k_ = nullptr;
};

using RingPair = std::pair<PolynomialRingZq, PolynomialRingZq>;
class RingLPR : public PKEncryptionScheme<RingLPR, RingPair, PolynomialRingZq, RingPair> {
  public:
    uint32_t n, q;
    double alpha;
    RingLPR(const Device &d, uint32_t n_, uint32_t q_, double alpha_, Seed seed = Seed::fresh())
        : n(n_), q(q_), alpha(alpha_), dev(&d), rng(d, seed), ring(d, n_, q_) { default_noise(rng, n, alpha * q); }
    static RingLPR default_(const Device &d) { return RingLPR(d, 16, 2399, 0.0039); }             // ring_lpr.rs:343-345
    static RingLPR secure128(const Device &d) { return RingLPR(d, 512, 92897729, 0.000005); }     // ring_lpr.rs:312-314
    void check_correctness() const {  // ring_lpr.rs:228-270
        if (n <= 1) throw MathError("n must be chosen bigger than 1.");
        if (n & (n - 1)) throw MathError("n is not a perfect power of 2, which is required for the correctness of this scheme.");
        if (alpha > 1.0 / (lpr_factor(n) * std::sqrt((double)n) * std::pow(std::log2((double)n), 3)))
            throw MathError("Correctness is not guaranteed as alpha >= 1 / (sqrt(n) * log^3 n)");
    }
    void check_security() const {  // ring_lpr.rs:294-306
        if (q * alpha < 2 * std::sqrt((double)n)) throw MathError("Security is not guaranteed as q * alpha < 2 * sqrt(n)");
    }
    uint32_t noise_width() const { return gaussian_width(alpha * q); }
    uint32_t mu_stride() const { return (n + 7) / 8; }
    PolynomialRingZq poly(std::vector<uint32_t> c) const { PolynomialRingZq p; p.n = n; p.q = q; p.c = std::move(c); return p; }

    std::pair<RingPair, PolynomialRingZq> gen() {  // ring_lpr.rs:370-395: b = a*s + e
        const uint32_t W = noise_width();
        std::vector<uint32_t> a = rng.uniform(1, n, q), s = rng.noise(2, n, q, W), e = rng.noise(4, n, q, W);
        DevBuf<uint32_t> da(*dev, a), ds(*dev, s), dc(*dev, n);
        dev->check(qf_polymul_dev(dev->ctx(), ring.get(), da.get(), ds.get(), 1, dc.get()));
        std::vector<uint32_t> b = dc.download();
        for (uint32_t i = 0; i < n; i++) b[i] = (uint32_t)(((uint64_t)b[i] + e[i]) % q);
        return {{poly(a), poly(b)}, poly(s)};
    }
    // ring_lpr.rs:422-461 for B messages given as packed bit rows (bit k of row i = bit k of |msg_i| mod 2^n)
    std::pair<DevBuf<uint32_t>, DevBuf<uint32_t>> enc_batch(const RingPair &pk, const std::vector<uint8_t> &mu_bits, uint64_t B,
                                                            const std::vector<uint32_t> &R, const std::vector<uint32_t> &E1, const std::vector<uint32_t> &E2) {
        if (mu_bits.size() != B * mu_stride() || R.size() != B * n || E1.size() != B * n || E2.size() != B * n) throw Panic("mismatching dimensions");
        std::vector<uint32_t> kp(pk.first.c);
        kp.insert(kp.end(), pk.second.c.begin(), pk.second.c.end());
        RingKey key(*dev, ring, kp, 2);
        DevBuf<uint32_t> dr(*dev, R), d1(*dev, E1), d2(*dev, E2), u(*dev, B * n), v(*dev, B * n);
        DevBuf<uint8_t> dmu(*dev, mu_bits);
        dev->check(qf_ringlpr_enc_dev(dev->ctx(), ring.get(), key.get(), dr.get(), d1.get(), d2.get(), dmu.get(), B, u.get(), v.get()));
        dev->sync();
        return {std::move(u), std::move(v)};
    }
    // message: a non-negative integer below 2^64 (the reference takes any Z and reduces |message| mod 2^n)
    RingPair enc(const RingPair &pk, int64_t message) {
        const uint32_t W = noise_width();
        uint64_t v = message < 0 ? (uint64_t)(-(message + 1)) + 1 : (uint64_t)message;
        std::vector<uint8_t> mu(mu_stride(), 0);
        for (uint32_t k = 0; k < 64 && k < n; k++) if ((v >> k) & 1) mu[k >> 3] |= (uint8_t)(1u << (k & 7));
        auto uv = enc_batch(pk, mu, 1, rng.noise(3, n, q, W), rng.noise(4, n, q, W), rng.noise(4, n, q, W));
        return {poly(uv.first.download()), poly(uv.second.download())};
    }
    std::vector<uint8_t> dec_batch(const PolynomialRingZq &sk, const uint32_t *u_dev, const uint32_t *v_dev, uint64_t B) {  // :489-508
        RingKey key(*dev, ring, sk.c, 1);
        DevBuf<uint8_t> out(*dev, B * mu_stride());
        dev->check(qf_ringlpr_dec_dev(dev->ctx(), ring.get(), key.get(), u_dev, v_dev, B, out.get()));
        return out.download();
    }
    int64_t dec(const PolynomialRingZq &sk, const RingPair &cipher) {  // low 63 bits of the message
        DevBuf<uint32_t> du(*dev, cipher.first.c), dv(*dev, cipher.second.c);
        std::vector<uint8_t> bits = dec_batch(sk, du.get(), dv.get(), 1);
        uint64_t m = 0;
        for (uint32_t k = 0; k < 63 && k < n; k++) if ((bits[k >> 3] >> (k & 7)) & 1) m |= 1ull << k;
        return (int64_t)m;
    }
    const Ring &get_ring() const { return ring; }
    Randomness &randomness() { return rng; }

  private:
    const Device *dev;
    Randomness rng;
    Ring ring;
};

// ---------------------------------------------------------------------------------------------
// RegevWithDiscreteGaussianRegularity / DualRegevWithDiscreteGaussianRegularity
// (regev_discrete_gauss.rs, dual_regev_discrete_gauss.rs): the same kernels as Regev / DualRegev;
// chi = D_{Z,alpha*q} and SampleD over Z^m (= m independent D_{Z,r} draws) come from
// qf_sample_discrete_gauss_dev.
// ---------------------------------------------------------------------------------------------
struct GaussRegularityParams {
    uint32_t n, m, q;
    double r, alpha;
    // regev_discrete_gauss.rs:161-199: q prime in [n^p/2, n^p], m = ceil(2(n+1) lg q) (lg = log10 there),
    // r = log2 m, alpha = 1/(sqrt(m) log2^2 m); "random prime" -> largest prime, next smaller while checks fail
    static GaussRegularityParams from_n(uint32_t n) {
        if (n <= 1) throw Panic("n must be chosen bigger than 1.");
        const int power = n == 2 ? 9 : n == 3 ? 8 : n <= 5 ? 7 : n <= 8 ? 6 : n <= 12 ? 5 : n <= 30 ? 4 : 3;
        const uint64_t ub = ipow(n, power), lb = (ub + 1) / 2;
        uint64_t upper = ub;
        for (;;) {
            const uint64_t qq = largest_prime_at_most(upper, lb);
            GaussRegularityParams p;
            p.n = n; p.q = (uint32_t)qq;
            p.m = (uint32_t)std::ceil(2.0 * (n + 1) * std::log10((double)qq));
            p.r = std::log2((double)p.m);
            p.alpha = 1.0 / (std::sqrt((double)p.m) * std::pow(std::log2((double)p.m), 2));
            try { p.check_correctness(); p.check_security(); return p; } catch (const MathError &) { upper = qq - 1; }
        }
    }
    void check_correctness() const {  // regev_discrete_gauss.rs:222-246
        if (n <= 1) throw MathError("n must be chosen bigger than 1.");
        if (q < 5 * r * m) throw MathError("Correctness is not guaranteed as q < 5rm, but q >= 5rm is required.");
        if (alpha > 1 / (r * std::sqrt((double)m) * std::sqrt(std::log2((double)n)))) throw MathError("Correctness is not guaranteed as alpha > 1/(r*sqrt(m)*w(sqrt(log n))");
    }
    void check_security() const {  // regev_discrete_gauss.rs:270-297
        if (q * alpha < n) throw MathError("Security is not guaranteed as q * alpha < n, but q * alpha >= n is required.");
        if (m < 2.0 * (n + 1) * std::log10((double)q)) throw MathError("Security is not guaranteed as m < 2(n + 1) lg (q)");
        if (r < std::sqrt(std::log2((double)m))) throw MathError("Security is not guaranteed as r < sqrt( log m )");
    }
};

struct GaussCipher { MatZq vec; uint32_t c = 0; };       // (u, c) resp. (p, c): (MatZq, Zq)
struct GaussPublicKey { MatZq a; MatZq v; };             // (A, p) resp. (A, u)

class RegevWithDiscreteGaussianRegularity
    : public PKEncryptionScheme<RegevWithDiscreteGaussianRegularity, GaussPublicKey, MatZq, GaussCipher> {
  public:
    GaussRegularityParams pp;
    RegevWithDiscreteGaussianRegularity(const Device &d, uint32_t n, uint32_t m, uint32_t q, double r, double alpha, Seed seed = Seed::fresh())
        : pp{n, m, q, r, alpha}, dev(&d), rng(d, seed) { if (q <= 1) throw Panic("modulus must be > 1"); }
    static RegevWithDiscreteGaussianRegularity default_(const Device &d) { return {d, 2, 16, 443, 4, 1.0 / 64}; }             // :329-338
    static RegevWithDiscreteGaussianRegularity secure128(const Device &d) { return {d, 350, 5248, 29892991, 12.357, 0.00009}; }  // :312-314
    static RegevWithDiscreteGaussianRegularity new_from_n(const Device &d, uint32_t n) {
        GaussRegularityParams p = GaussRegularityParams::from_n(n);
        return {d, p.n, p.m, p.q, p.r, p.alpha};
    }
    // regev_discrete_gauss.rs:369-393: s <- Z_q^n, A <- Z_q^{n x m}, x <- chi^m, p = A^t s + x
    std::pair<GaussPublicKey, MatZq> gen() {
        MatZq sk(pp.n, 1, pp.q), A(pp.n, pp.m, pp.q), p(pp.m, 1, pp.q);
        sk.e = rng.uniform(2, pp.n, pp.q);
        A.e = rng.uniform(1, (uint64_t)pp.n * pp.m, pp.q);
        std::vector<uint32_t> x = gauss(4, pp.m, pp.alpha * pp.q);
        { DevKey ka(*dev, A); dev->check(qf_vecmat_modq_host(dev->ctx(), ka.get(), sk.e.data(), p.e.data())); }
        for (uint32_t j = 0; j < pp.m; j++) p.e[j] = (uint32_t)(((uint64_t)p.e[j] + x[j]) % pp.q);
        return {GaussPublicKey{A, p}, sk};
    }
    // regev_discrete_gauss.rs:413-430: e <- SampleD(Z^m, r); u = A e, c = p^t e + msg floor(q/2)
    GaussCipher enc(const GaussPublicKey &pk, int64_t message) {
        MatZq K = stacked(pk);
        std::vector<uint32_t> e = gauss(3, pp.m, pp.r), y(pp.n + 1);
        qf_key *k = dev->cached_key(K);
        dev->check(qf_matvec_modq_host(dev->ctx(), k, e.data(), y.data()));
        GaussCipher out{MatZq(pp.n, 1, pp.q), (uint32_t)(((uint64_t)y[pp.n] + (uint64_t)message_bit(message) * (pp.q / 2)) % pp.q)};
        std::copy(y.begin(), y.begin() + pp.n, out.vec.e.begin());
        return out;
    }
    // regev_discrete_gauss.rs:451-462: c - <s, u>
    int64_t dec(const MatZq &sk, const GaussCipher &cipher) {
        if (sk.rows != pp.n || cipher.vec.rows != pp.n) throw Panic("mismatching matrix dimensions");
        std::vector<uint32_t> w(pp.n), row(cipher.vec.e);
        for (uint32_t i = 0; i < pp.n; i++) w[i] = (pp.q - sk.e[i] % pp.q) % pp.q;
        row.push_back(cipher.c % pp.q);
        DevBuf<uint32_t> dc(*dev, row);
        return decode_batch(*dev, w, pp.q, dc.get(), 1)[0];
    }
    // B encryptions on kernel (a) with E = B x m SampleD rows (device-drawn): rows (u_b ; c_b)
    DevBuf<uint32_t> enc_batch(const GaussPublicKey &pk, const std::vector<uint8_t> &msg_bits) {
        MatZq K = stacked(pk);
        const uint64_t B = msg_bits.size();
        DevBuf<uint32_t> E(*dev, B * pp.m), out(*dev, B * (pp.n + 1));
        rng.gauss_dev(3, B * pp.m, pp.q, pp.n, pp.r, E.get());
        qf_key *k = dev->cached_key(K);
        DevBuf<uint8_t> dm(*dev, msg_bits);
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, E.get(), QF_X_U32, (size_t)pp.m * 4, 0, B, nullptr, dm.get(), out.get()));
        return out;
    }
    std::vector<uint8_t> dec_batch(const MatZq &sk, const uint32_t *C_dev, uint64_t B) {
        std::vector<uint32_t> w(pp.n);
        for (uint32_t i = 0; i < pp.n; i++) w[i] = (pp.q - sk.e[i] % pp.q) % pp.q;
        return decode_batch(*dev, w, pp.q, C_dev, B);
    }
    Randomness &randomness() { return rng; }

  private:
    std::vector<uint32_t> gauss(uint32_t stream, uint64_t count, double s) {
        return rng.gauss(stream, count, pp.q, pp.n, s);
    }
    MatZq stacked(const GaussPublicKey &pk) const {   // K = [A ; p^t]
        if (pk.a.q != pp.q || pk.v.q != pp.q) throw Panic("mismatching modulus between scheme and key");
        if (pk.a.rows != pp.n || pk.a.cols != pp.m || pk.v.rows != pp.m || pk.v.cols != 1) throw Panic("mismatching matrix dimensions");
        MatZq K(pp.n + 1, pp.m, pp.q);
        std::copy(pk.a.e.begin(), pk.a.e.end(), K.e.begin());
        std::copy(pk.v.e.begin(), pk.v.e.end(), K.e.begin() + (size_t)pp.n * pp.m);
        return K;
    }
    const Device *dev;
    Randomness rng;
};

class DualRegevWithDiscreteGaussianRegularity
    : public PKEncryptionScheme<DualRegevWithDiscreteGaussianRegularity, GaussPublicKey, MatZq, GaussCipher> {
  public:
    GaussRegularityParams pp;
    DualRegevWithDiscreteGaussianRegularity(const Device &d, uint32_t n, uint32_t m, uint32_t q, double r, double alpha, Seed seed = Seed::fresh())
        : pp{n, m, q, r, alpha}, dev(&d), rng(d, seed) { if (q <= 1) throw Panic("modulus must be > 1"); }
    static DualRegevWithDiscreteGaussianRegularity default_(const Device &d) { return {d, 2, 16, 443, 4, 1.0 / 64}; }
    static DualRegevWithDiscreteGaussianRegularity secure128(const Device &d) { return {d, 350, 5248, 29892991, 12.357, 0.00009}; }
    static DualRegevWithDiscreteGaussianRegularity new_from_n(const Device &d, uint32_t n) {
        GaussRegularityParams p = GaussRegularityParams::from_n(n);
        return {d, p.n, p.m, p.q, p.r, p.alpha};
    }
    // dual_regev_discrete_gauss.rs:369-383: A uniform, e <- SampleD(Z^m, r), u = A e; sk = e
    std::pair<GaussPublicKey, MatZq> gen() {
        MatZq A(pp.n, pp.m, pp.q), u(pp.n, 1, pp.q), sk(pp.m, 1, pp.q);
        A.e = rng.uniform(1, (uint64_t)pp.n * pp.m, pp.q);
        sk.e = gauss(2, pp.m, pp.r);
        { DevKey ka(*dev, A); dev->check(qf_matvec_modq_host(dev->ctx(), ka.get(), sk.e.data(), u.e.data())); }
        return {GaussPublicKey{A, u}, sk};
    }
    // dual_regev_discrete_gauss.rs:404-431: p = A^t s + x, c = u^t s + x' + msg floor(q/2)
    GaussCipher enc(const GaussPublicKey &pk, int64_t message) {
        MatZq K = joined(pk);
        std::vector<uint32_t> s = rng.uniform(3, pp.n, pp.q), x = gauss(4, pp.m + 1, pp.alpha * pp.q), y(pp.m + 1);
        qf_key *k = dev->cached_key(K);
        dev->check(qf_vecmat_modq_host(dev->ctx(), k, s.data(), y.data()));
        for (uint32_t j = 0; j <= pp.m; j++) y[j] = (uint32_t)(((uint64_t)y[j] + x[j]) % pp.q);
        GaussCipher out{MatZq(pp.m, 1, pp.q), (uint32_t)(((uint64_t)y[pp.m] + (uint64_t)message_bit(message) * (pp.q / 2)) % pp.q)};
        std::copy(y.begin(), y.begin() + pp.m, out.vec.e.begin());
        return out;
    }
    // dual_regev_discrete_gauss.rs:455-466: c - <e, p>
    int64_t dec(const MatZq &sk, const GaussCipher &cipher) {
        if (sk.rows != pp.m || cipher.vec.rows != pp.m) throw Panic("mismatching matrix dimensions");
        std::vector<uint32_t> w(pp.m), row(cipher.vec.e);
        for (uint32_t i = 0; i < pp.m; i++) w[i] = (pp.q - sk.e[i] % pp.q) % pp.q;
        row.push_back(cipher.c % pp.q);
        DevBuf<uint32_t> dc(*dev, row);
        return decode_batch(*dev, w, pp.q, dc.get(), 1)[0];
    }
    Randomness &randomness() { return rng; }

  private:
    std::vector<uint32_t> gauss(uint32_t stream, uint64_t count, double s) {
        return rng.gauss(stream, count, pp.q, pp.n, s);
    }
    MatZq joined(const GaussPublicKey &pk) const {   // K = [A | u]
        if (pk.a.q != pp.q || pk.v.q != pp.q) throw Panic("mismatching modulus between scheme and key");
        if (pk.a.rows != pp.n || pk.a.cols != pp.m || pk.v.rows != pp.n || pk.v.cols != 1) throw Panic("mismatching matrix dimensions");
        MatZq K(pp.n, pp.m + 1, pp.q);
        for (uint32_t i = 0; i < pp.n; i++) {
            std::copy(pk.a.e.begin() + (size_t)i * pp.m, pk.a.e.begin() + (size_t)(i + 1) * pp.m, K.e.begin() + (size_t)i * (pp.m + 1));
            K.e[(size_t)i * (pp.m + 1) + pp.m] = pk.v.e[i];
        }
        return K;
    }
    const Device *dev;
    Randomness rng;
};

// ---------------------------------------------------------------------------------------------
// classical GPV PSF, verify half (primitive/psf/gpv.rs:189-223): f_a = A*sigma, check_domain
// ---------------------------------------------------------------------------------------------
class PSFGPV {
  public:
    uint32_t n, q, k, m_bar;
    double s;
    PSFGPV(const Device &d, uint32_t n_, uint32_t q_, double s_) : n(n_), q(q_), s(s_), dev(&d) {   // gadget_parameters.rs:86-108
        if (n < 1) throw Panic("n must be at least 1");
        k = log2_ceil(q);
        const uint32_t ln = log2_ceil(n);
        m_bar = n * k + ln * ln;
    }
    uint32_t m() const { return n * k + m_bar; }
    bool check_domain(const MatZ &sigma) const {   // gpv.rs:218-223 (exact: the norm in 128-bit integers)
        if (sigma.cols != 1 || sigma.rows != m() || sigma.e.size() != m()) return false;
        unsigned __int128 norm = 0;
        for (int64_t v : sigma.e) norm += (unsigned __int128)((__int128)v * v);
        return (long double)norm <= (long double)s * s * m();
    }
    MatZq f_a(const MatZq &a, const MatZ &sigma) const {   // gpv.rs:189-192
        if (!check_domain(sigma)) throw Panic("sigma is not in the domain");
        if (a.q != q || a.rows != n || a.cols != m()) throw Panic("mismatching matrix dimensions");
        std::vector<uint32_t> x(m());
        for (uint32_t i = 0; i < m(); i++) x[i] = (uint32_t)(((sigma.e[i] % (int64_t)q) + q) % q);
        MatZq out(n, 1, q);
        qf_key *key = dev->cached_key(a);
        dev->check(qf_matvec_modq_host(dev->ctx(), key, x.data(), out.e.data()));
        return out;
    }

  private:
    const Device *dev;
};

// ---------------------------------------------------------------------------------------------
// SISHash (hash/sis.rs)
// ---------------------------------------------------------------------------------------------
class SISHash {
  public:
    MatZq key;
    SISHash(const Device &d, MatZq k) : key(std::move(k)), dev(&d) {}
    static SISHash gen(const Device &d, long n, uint32_t m, uint32_t q, Seed seed = Seed::fresh()) {  // sis.rs:79-93
        if (n < 1) throw MathError("n must be chosen bigger than 0.");
        Randomness rng(d, seed);
        MatZq A((uint32_t)n, m, q);
        A.e = rng.uniform(1, (uint64_t)n * m, q);
        return SISHash(d, std::move(A));
    }
    void check_security() const {  // sis.rs:114-136
        const double mbar = std::ceil(key.rows * std::log2((double)key.q));
        if (key.cols < mbar) throw MathError("m was chosen smaller than n log q");
        if (key.q <= std::ceil(std::sqrt(mbar))) throw MathError("q was chosen smaller than ceil(sqrt(n log q))");
    }
    MatZq hash(const MatZq &value) const {  // sis.rs:161-167
        if (!value.is_column_vector()) throw Panic("The hashed value has to be a column vector!");
        if (value.q != key.q) throw Panic("mismatching modulus");
        if (value.rows != key.cols) throw Panic("mismatching matrix dimensions");
        MatZq out(key.rows, 1, key.q);
        qf_key *k = dev->cached_key(key);
        dev->check(qf_matvec_modq_host(dev->ctx(), k, value.e.data(), out.e.data()));
        return out;
    }
    // B binary inputs (packed rows of bit_stride(m) bytes) on kernel (a): B x n digests on the device
    DevBuf<uint32_t> hash_batch_bits(const std::vector<uint8_t> &x_bits, uint64_t B) const {
        if (x_bits.size() != B * bit_stride(key.cols)) throw Panic("mismatching matrix dimensions");
        qf_key *k = dev->cached_key(key);
        DevBuf<uint8_t> dx(*dev, x_bits);
        DevBuf<uint32_t> out(*dev, B * key.rows);
        dev->check(qf_matmul_modq_dev(dev->ctx(), k, dx.get(), QF_X_BITS, bit_stride(key.cols), 1, B, nullptr, nullptr, out.get()));
        return out;
    }

  private:
    const Device *dev;
};

// ---------------------------------------------------------------------------------------------
// GPV-ring PSF verify half + FDH::vfy (primitive/psf/gpv_ring.rs:235-275, signature/fdh.rs:115-123,
// signature/fdh/gpv_ring.rs:65-87).  trap_gen / samp_d / samp_p / sign are out of scope.
// ---------------------------------------------------------------------------------------------
class FdhGpvRing {
  public:
    uint32_t n, q, k;  // k = ceil(log2 q); the parity-check row has k+2 polynomials (gadget_parameters.rs:167-188)
    double s;
    FdhGpvRing(const Device &d, uint32_t n_, uint32_t q_, double s_) : n(n_), q(q_), k(log2_ceil(q_)), s(s_), dev(&d), ring(d, n_, q_) {}
    static FdhGpvRing init_gpv_ring(const Device &d, uint32_t n, uint32_t q, double s) { return FdhGpvRing(d, n, q, s); }
    uint32_t mbar() const { return k + 2; }
    uint64_t norm_bound() const { return (uint64_t)std::floor(s * s * (double)n * mbar()); }  // gpv_ring.rs:273-274
    // check_domain for B signatures (sigma: B x (k+2) x n signed coefficients)
    std::vector<uint8_t> check_domain_batch(const std::vector<int32_t> &sigma, uint64_t B) const {
        std::vector<uint32_t> zero_key((size_t)mbar() * n, 0), zero_u((size_t)B * n, 0);
        return verify_core(zero_key, sigma, zero_u, B, nullptr);  // a = 0 => t = 0 = u: ok == in_domain
    }
    // f_a: t_b = sum_j a_j * (sigma_b[j] mod q); asserts the domain like gpv_ring.rs:236
    std::vector<uint32_t> f_a_batch(const std::vector<uint32_t> &a, const std::vector<int32_t> &sigma, uint64_t B) const {
        for (uint8_t ok : check_domain_batch(sigma, B)) if (!ok) throw Panic("sigma is not in the domain");
        std::vector<uint32_t> t, zero_u((size_t)B * n, 0);
        verify_core(a, sigma, zero_u, B, &t);
        return t;
    }
    // FDH::vfy for B (message, signature) pairs: the hash runs on the device as well
    std::vector<uint8_t> vfy_batch(const std::vector<std::string> &messages, const std::vector<int32_t> &sigma, const std::vector<uint32_t> &pk) const {
        const uint64_t B = messages.size();
        std::vector<uint8_t> blob;
        std::vector<uint64_t> offs(B + 1, 0);
        for (uint64_t b = 0; b < B; b++) { blob.insert(blob.end(), messages[b].begin(), messages[b].end()); offs[b + 1] = blob.size(); }
        if (blob.empty()) blob.push_back(0);
        DevBuf<uint8_t> dblob(*dev, blob);
        DevBuf<uint64_t> doffs(*dev, offs);
        DevBuf<uint32_t> du(*dev, B * n);
        dev->check(qf_hash_to_mat_zq_sha256_dev(dev->ctx(), dblob.get(), doffs.get(), B, n, 1, q, du.get()));
        return verify_dev(pk, sigma, du.get(), B, nullptr);
    }
    bool vfy(const std::string &m, const MatPolyOverZ &sigma, const std::vector<uint32_t> &pk) const {
        if (sigma.rows != mbar() || sigma.n != n) return false;
        return vfy_batch({m}, sigma.c, pk)[0] != 0;
    }

  private:
    std::vector<uint8_t> verify_core(const std::vector<uint32_t> &a, const std::vector<int32_t> &sigma, const std::vector<uint32_t> &u, uint64_t B, std::vector<uint32_t> *t) const {
        DevBuf<uint32_t> du(*dev, u);
        return verify_dev(a, sigma, du.get(), B, t);
    }
    std::vector<uint8_t> verify_dev(const std::vector<uint32_t> &a, const std::vector<int32_t> &sigma, const uint32_t *u_dev, uint64_t B, std::vector<uint32_t> *t) const {
        if (a.size() != (size_t)mbar() * n || sigma.size() != (size_t)B * mbar() * n) throw Panic("mismatching dimensions");
        RingKey key(*dev, ring, a, mbar());
        DevBuf<int32_t> ds(*dev, sigma);
        DevBuf<uint32_t> dt(*dev, t ? B * n : 0);
        DevBuf<uint8_t> ok(*dev, B);
        dev->check(qf_fdh_ring_verify_dev(dev->ctx(), ring.get(), key.get(), ds.get(), u_dev, norm_bound(), B, t ? dt.get() : nullptr, ok.get()));
        if (t) *t = dt.download();
        return ok.download();
    }
    const Device *dev;
    Ring ring;
};

// utils/rotation_matrix.rs:45-67: column j of rot_minus(a) holds the coefficients of a * X^j mod X^n+1
inline std::vector<int64_t> rot_minus(const std::vector<int64_t> &a) {
    const size_t n = a.size();
    std::vector<int64_t> out(n * n);
    for (size_t j = 0; j < n; j++)
        for (size_t i = 0; i < n; i++) out[i * n + j] = i >= j ? a[i - j] : -a[n + i - j];
    return out;
}

}  // namespace qfall
