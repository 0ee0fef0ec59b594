// Host-layer tests in the reference's own style (randomised round trips, shape checks, panics;
// SURVEY section 4), written against include/qfall_crypto.hpp, plus bit-exact comparisons with
// the CPU oracle (oracle/qf_oracle.c) on the same seeded inputs.  Built by __graft_entry__.build()
// into tests/cpp/test_host; run on the GPU box by tests/test_gpu_cpp_host.py.
#include <climits>
#include <cstdio>
#include <cstring>
#include <functional>
#include "../../include/qfall_crypto.hpp"

extern "C" {  // oracle (test infrastructure)
void qfo_matvec(const uint32_t *A, uint32_t rows, uint32_t cols, uint32_t q, const uint32_t *x, uint32_t *y);
void qfo_vecmat(const uint32_t *A, uint32_t rows, uint32_t cols, uint32_t q, const uint32_t *x, uint32_t *y);
void qfo_matmul_batch(const uint32_t *A, uint32_t rows, uint32_t cols, uint32_t q, const void *X, int xkind, size_t xstride_bytes,
                      uint64_t B, const uint32_t *addend, const uint8_t *msg, uint32_t *out);
void qfo_dot_decode_batch(const uint32_t *s, uint32_t n, uint32_t q, const uint32_t *C, uint64_t B, uint8_t *bits);
void qfo_sample_bits(uint64_t seed, uint32_t stream, uint64_t first_row, uint64_t rows, uint32_t cols, uint32_t stride_bytes, uint8_t *out);
void qfo_sample_noise(uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t W, uint32_t *out);
void qfo_sample_uniform(uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t *out);
int qfo_sample_discrete_gauss(uint64_t seed, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t n, double s, uint32_t *out);
int qfo_sample_discrete_gauss_keyed(const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t n,
                                    double s, uint32_t *out);
void qfo_sample_uniform_keyed(const uint8_t *key32, uint64_t nonce, uint32_t stream, uint64_t first, uint64_t count, uint32_t q, uint32_t *out);
void qfo_polymul_negacyclic(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t q, uint32_t *c);
void qfo_ringlpr_enc_batch(const uint32_t *a, const uint32_t *bpoly, uint32_t n, uint32_t q, const uint32_t *r, const uint32_t *e1,
                           const uint32_t *e2, const uint8_t *mu_bits, uint64_t B, uint32_t *u, uint32_t *v);
void qfo_fdh_ring_verify_batch(const uint32_t *A, uint32_t mbar, uint32_t n, uint32_t q, const int32_t *sigma, const uint32_t *u,
                               uint64_t norm_bound, uint64_t B, uint32_t *t_out, uint8_t *ok);
void qfo_hash_to_mat_zq(const char *str, size_t len, uint32_t rows, uint32_t cols, uint32_t q, uint32_t *out);
}

using namespace qfall;
static int failures = 0, checks = 0;
#define CHECK(cond) do { checks++; if (!(cond)) { failures++; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)
template <class E, class F> static bool throws(F f) { try { f(); } catch (const E &) { return true; } catch (...) { return false; } return false; }
static void run(const char *name, const std::function<void()> &f) {
    int before = failures;
    try { f(); } catch (const std::exception &e) { failures++; std::printf("FAIL %s threw: %s\n", name, e.what()); }
    std::printf("%-44s %s\n", name, failures == before ? "ok" : "FAILED");
    std::fflush(stdout);
}

int main() {
    Device dev(0);

    // ---- Regev: regev.rs:547-595 cycle_zero/one at default and new_from_n(50) ----------------
    run("regev_cycles", [&] {
        for (int which = 0; which < 2; which++) {
            Regev r = which ? Regev::new_from_n(dev, 50) : Regev::default_(dev);
            if (which) { CHECK(r.n == 50 && r.q == 124991 && r.m == 867); r.check_correctness(); r.check_security(); }
            auto [pk, sk] = r.gen();
            CHECK(pk.rows == r.n + 1 && pk.cols == r.m && sk.rows == r.n);
            for (int64_t msg : {0, 1, 0, 1, 1}) CHECK(r.dec(sk, r.enc(pk, msg)) == msg);
        }
    });
    run("regev_modulo_two", [&] {  // regev.rs:598-612
        Regev r = Regev::default_(dev);
        auto [pk, sk] = r.gen();
        CHECK(r.dec(sk, r.enc(pk, 3)) == 1);
        CHECK(r.dec(sk, r.enc(pk, -2)) == 0);
        CHECK(r.dec(sk, r.enc(pk, LLONG_MAX)) == 1);
        CHECK(r.dec(sk, r.enc(pk, LLONG_MIN)) == 0);
    });
    run("regev_multi_bit", [&] {   // regev.rs:622-669, pk_encryption.rs:130-166
        Regev r = Regev::new_from_n(dev, 20);
        auto [pk, sk] = r.gen();
        for (int64_t msg : {0LL, 1LL, 15LL, 212LL, 65535LL, -13LL}) {
            auto c = r.enc_multiple_bits(pk, msg);
            CHECK((int64_t)r.dec_multiple_bits(sk, c) == (msg < 0 ? -msg : msg));
        }
    });
    run("regev_enc_matches_oracle", [&] {
        Regev r = Regev::new_from_n(dev, 50);
        auto [pk, sk] = r.gen();
        std::vector<uint32_t> x = r.randomness().bits(3, r.m), ref(r.n + 1);
        MatZq c = r.enc(pk, 1, &x);
        qfo_matvec(pk.e.data(), r.n + 1, r.m, r.q, x.data(), ref.data());
        ref[r.n] = (uint32_t)(((uint64_t)ref[r.n] + r.q / 2) % r.q);
        CHECK(c.e == ref);
        // batch: 300 messages through kernel (a), bit-exact with the oracle, decrypting to the messages
        const uint64_t B = 300;
        std::vector<uint8_t> xb(B * bit_stride(r.m)), msg(B), bits_ref(B);
        qfo_sample_bits(MASTER_SEED, 3, 1000, B, r.m, bit_stride(r.m), xb.data());
        for (uint64_t b = 0; b < B; b++) msg[b] = (uint8_t)((b * 7 + 3) & 1);
        DevBuf<uint32_t> C = r.enc_batch(pk, msg, xb);
        std::vector<uint32_t> Ch = C.download(), Cref(B * (r.n + 1));
        qfo_matmul_batch(pk.e.data(), r.n + 1, r.m, r.q, xb.data(), 0, bit_stride(r.m), B, nullptr, msg.data(), Cref.data());
        CHECK(Ch == Cref);
        std::vector<uint8_t> bits = r.dec_batch(sk, C.get(), B);
        qfo_dot_decode_batch(sk.e.data(), r.n, r.q, Cref.data(), B, bits_ref.data());
        CHECK(bits == bits_ref && bits == msg);
        // x drawn on the device, k-tile-major (QF_X_BITS_KT): the same ciphertexts as the oracle computes from the rows
        // the generator gives for that position of stream 3
        Regev rs(dev, r.n, r.m, r.q, r.alpha, Seed(MASTER_SEED));
        const uint64_t pos = rs.randomness().take(3, 0);
        DevBuf<uint32_t> C2 = rs.enc_batch(pk, msg);
        qfo_sample_bits(MASTER_SEED, 3, pos, B, r.m, bit_stride(r.m), xb.data());
        qfo_matmul_batch(pk.e.data(), r.n + 1, r.m, r.q, xb.data(), 0, bit_stride(r.m), B, nullptr, msg.data(), Cref.data());
        CHECK(C2.download() == Cref);
        CHECK(rs.dec_batch(sk, C2.get(), B) == msg);
    });
    run("regev_panics", [&] {
        Regev r = Regev::default_(dev);
        auto [pk, sk] = r.gen();
        MatZq bad(pk.rows, pk.cols - 1, pk.q);
        CHECK(throws<Panic>([&] { r.enc(bad, 1); }));
        MatZq badq = pk; badq.q = 1429;
        CHECK(throws<Panic>([&] { r.enc(badq, 1); }));
        CHECK(throws<Panic>([&] { Regev::new_from_n(dev, 9); }));
        CHECK(throws<MathError>([&] { Regev(dev, 13, 154, 1427, 0.5).check_correctness(); }));
    });

    // ---- DualRegev: dual_regev.rs:545-593 ------------------------------------------------------
    run("dual_regev_cycles_and_oracle", [&] {
        DualRegev d = DualRegev::new_from_n(dev, 20);
        auto [pk, sk] = d.gen();
        CHECK(pk.rows == d.n && pk.cols == d.m + 1 && sk.e.size() == d.m);
        for (int64_t msg : {0, 1, 1, 0}) CHECK(d.dec(sk, d.enc(pk, msg)) == msg);
        std::vector<uint32_t> s = d.randomness().uniform(3, d.n, d.q), e = d.randomness().noise(4, d.m + 1, d.q, d.noise_width()), ref(d.m + 1);
        MatZq c = d.enc(pk, 1, &s, &e);
        qfo_vecmat(pk.e.data(), d.n, d.m + 1, d.q, s.data(), ref.data());
        for (uint32_t j = 0; j <= d.m; j++) ref[j] = (uint32_t)(((uint64_t)ref[j] + e[j]) % d.q);
        ref[d.m] = (uint32_t)(((uint64_t)ref[d.m] + d.q / 2) % d.q);
        CHECK(c.e == ref);
    });

    // ---- LPR: lpr.rs:564-616 -------------------------------------------------------------------
    run("lpr_cycles_and_oracle", [&] {
        for (int which = 0; which < 2; which++) {
            LPR l = which ? LPR::new_from_n(dev, 50) : LPR::default_(dev);
            auto [pk, sk] = l.gen();
            for (int64_t msg : {0, 1, 1, 0, 1}) CHECK(l.dec(sk, l.enc(pk, msg)) == msg);
            if (!which) continue;
            const uint64_t B = 130;
            const uint32_t W = l.noise_width(), n = l.n, q = l.q;
            std::vector<uint32_t> R(B * n), E(B * (n + 1)), Cref(B * (n + 1));
            std::vector<uint8_t> msg(B), bits_ref(B);
            qfo_sample_noise(MASTER_SEED, 3, 5000, B * n, q, W, R.data());
            qfo_sample_noise(MASTER_SEED, 4, 9000, B * (n + 1), q, W, E.data());
            for (uint64_t b = 0; b < B; b++) msg[b] = (uint8_t)((b / 3) & 1);
            DevBuf<uint32_t> C = l.enc_batch(pk, msg, R, E, l.noise_digits());
            qfo_matmul_batch(pk.e.data(), n + 1, n, q, R.data(), 1, (size_t)n * 4, B, E.data(), msg.data(), Cref.data());
            CHECK(C.download() == Cref);
            std::vector<uint8_t> bits = l.dec_batch(sk, C.get(), B);
            qfo_dot_decode_batch(sk.e.data(), n, q, Cref.data(), B, bits_ref.data());
            CHECK(bits == bits_ref && bits == msg);
        }
    });

    // ---- RingLPR: ring_lpr.rs:608-667 ----------------------------------------------------------
    run("ring_lpr_cycles", [&] {
        RingLPR r = RingLPR::default_(dev);   // n = 16, q = 2399: not NTT friendly -> exact schoolbook kernel
        CHECK(!r.get_ring().uses_ntt());
        auto [pk, sk] = r.gen();
        for (int64_t msg : {0LL, 1LL, 2LL, 15LL, 212LL, 65535LL, 255LL, 13LL}) CHECK(r.dec(sk, r.enc(pk, msg)) == msg);
        CHECK(r.dec(sk, r.enc(pk, 65536)) == 0);   // ring_lpr.rs:655-667: |message| mod 2^n
        CHECK(throws<MathError>([&] { RingLPR(dev, 16, 2399, 0.5).check_correctness(); }));
    });
    run("ring_lpr_n1024_matches_oracle", [&] {
        const uint32_t n = 1024, q = 1073479681, W = 36;
        RingLPR r(dev, n, q, 1.0 / 64 / (5 * 32.0 * 1000));
        CHECK(r.get_ring().uses_ntt());
        const uint64_t B = 6;
        std::vector<uint32_t> a(n), s(n), e(n), b(n), R(B * n), E1(B * n), E2(B * n), uref(B * n), vref(B * n);
        qfo_sample_uniform(MASTER_SEED, 1, 0, n, q, a.data());
        qfo_sample_noise(MASTER_SEED, 2, 0, n, q, W, s.data());
        qfo_sample_noise(MASTER_SEED, 4, 0, n, q, W, e.data());
        qfo_polymul_negacyclic(a.data(), s.data(), n, q, b.data());
        for (uint32_t i = 0; i < n; i++) b[i] = (uint32_t)(((uint64_t)b[i] + e[i]) % q);
        qfo_sample_noise(MASTER_SEED, 3, 0, B * n, q, W, R.data());
        qfo_sample_noise(MASTER_SEED, 4, n, B * n, q, W, E1.data());
        qfo_sample_noise(MASTER_SEED, 4, n + B * n, B * n, q, W, E2.data());
        std::vector<uint8_t> mu(B * r.mu_stride());
        qfo_sample_bits(MASTER_SEED, 5, 0, B, n, r.mu_stride(), mu.data());
        RingPair pk{r.poly(a), r.poly(b)};
        auto uv = r.enc_batch(pk, mu, B, R, E1, E2);
        qfo_ringlpr_enc_batch(a.data(), b.data(), n, q, R.data(), E1.data(), E2.data(), mu.data(), B, uref.data(), vref.data());
        CHECK(uv.first.download() == uref && uv.second.download() == vref);
        CHECK(r.dec_batch(r.poly(s), uv.first.get(), uv.second.get(), B) == mu);   // round trip of all n bits
    });

    // ---- SISHash: sis.rs:237-278 ---------------------------------------------------------------
    run("sis_hash", [&] {
        SISHash h = SISHash::gen(dev, 5, 60, 257);
        h.check_security();
        MatZq v(60, 1, 257), ref(5, 1, 257);
        qfo_sample_uniform(MASTER_SEED, 6, 0, 60, 257, v.e.data());
        MatZq d = h.hash(v);
        qfo_matvec(h.key.e.data(), 5, 60, 257, v.e.data(), ref.e.data());
        CHECK(d == ref && d.rows == 5 && d.cols == 1);
        CHECK(throws<Panic>([&] { h.hash(MatZq(1, 60, 257)); }));     // not a column vector
        CHECK(throws<Panic>([&] { h.hash(MatZq(60, 1, 259)); }));     // modulus mismatch
        CHECK(throws<Panic>([&] { h.hash(MatZq(59, 1, 257)); }));     // dimension mismatch
        CHECK(throws<MathError>([&] { SISHash::gen(dev, 0, 60, 257); }));
        CHECK(throws<MathError>([&] { SISHash::gen(dev, 5, 10, 257).check_security(); }));
        CHECK(MatZq(2, 2, 7).to_string() == "[[0, 0],[0, 0]] mod 7");
    });

    // ---- FDH over the GPV ring PSF: verify half ---------------------------------------------------
    run("fdh_ring_verify", [&] {
        const uint32_t n = 1024, q = 1073479681;
        FdhGpvRing f = FdhGpvRing::init_gpv_ring(dev, n, q, 1000.0);
        const uint32_t mbar = f.mbar();
        CHECK(mbar == 32);
        const uint64_t B = 4;
        std::vector<uint32_t> a((size_t)mbar * n), tmp((size_t)B * mbar * n);
        qfo_sample_uniform(MASTER_SEED, 1, 0, (uint64_t)mbar * n, q, a.data());
        for (uint32_t i = 0; i < n; i++) a[i] = i == 0;   // A_0 = 1 (gadget_ring.rs:71-77)
        qfo_sample_uniform(MASTER_SEED, 7, 0, B * mbar * n, 201, tmp.data());
        std::vector<int32_t> sigma(tmp.size());
        for (size_t i = 0; i < tmp.size(); i++) sigma[i] = (int32_t)tmp[i] - 100;
        std::vector<uint32_t> t = f.f_a_batch(a, sigma, B), tref(B * n), zero(B * n, 0);
        std::vector<uint8_t> okref(B);
        qfo_fdh_ring_verify_batch(a.data(), mbar, n, q, sigma.data(), zero.data(), f.norm_bound(), B, tref.data(), okref.data());
        CHECK(t == tref);
        // the string hash on the device equals the reference restatement (sha256.rs:110-133, KAT-pinned)
        std::vector<std::string> msgs = {"Hello World!", "", "qfall", std::string(200, 'x')};
        std::vector<uint32_t> href(n);
        std::vector<uint8_t> ok = f.vfy_batch(msgs, sigma, a);
        for (uint64_t b = 0; b < B; b++) CHECK(ok[b] == 0);   // a random sigma is no preimage of H(m)
        // out-of-domain signature: one huge coefficient
        std::vector<int32_t> big = sigma;
        big[5] = 2000000000;
        CHECK(f.check_domain_batch(big, B)[0] == 0 && f.check_domain_batch(sigma, B)[0] == 1);
        CHECK(throws<Panic>([&] { f.f_a_batch(a, big, B); }));
        MatPolyOverZ wrong; wrong.rows = mbar - 1; wrong.n = n; wrong.c.assign((size_t)(mbar - 1) * n, 0);
        CHECK(!f.vfy("m", wrong, a));
    });
    run("hash_to_ring_matches_oracle", [&] {
        const uint32_t n = 64, q = 12289;
        std::vector<std::string> msgs = {"Hello World!", "", std::string(119, 'k'), "qfall"};
        std::vector<uint8_t> blob; std::vector<uint64_t> offs(1, 0);
        for (auto &m : msgs) { blob.insert(blob.end(), m.begin(), m.end()); offs.push_back(blob.size()); }
        DevBuf<uint8_t> db(dev, blob); DevBuf<uint64_t> dof(dev, offs); DevBuf<uint32_t> out(dev, msgs.size() * n);
        dev.check(qf_hash_to_mat_zq_sha256_dev(dev.ctx(), db.get(), dof.get(), msgs.size(), n, 1, q, out.get()));
        std::vector<uint32_t> got = out.download(), ref(n);
        for (size_t b = 0; b < msgs.size(); b++) {
            qfo_hash_to_mat_zq(msgs[b].data(), msgs[b].size(), n, 1, q, ref.data());
            CHECK(std::equal(ref.begin(), ref.end(), got.begin() + b * n));
        }
    });
    run("discrete_gauss_noise_model", [&] {  // chi = D_{Z,alpha*q} (regev.rs:356-363, lpr.rs:342-360)
        Randomness rng(dev, 77);
        rng.use_discrete_gauss(1024, 6711.0);
        std::vector<uint32_t> got = rng.noise(4, 5000, 1073741789u, 0), ref(5000);
        CHECK(qfo_sample_discrete_gauss(77, 4, 0, 5000, 1073741789u, 1024, 6711.0, ref.data()) == 0);
        CHECK(got == ref);
        LPR l = LPR::new_from_n(dev, 64);
        l.randomness().use_discrete_gauss(l.n, l.alpha * l.q);
        auto [pk, sk] = l.gen();
        for (int64_t msg : {0, 1, 1, 0, 1}) CHECK(l.dec(sk, l.enc(pk, msg)) == msg);
        Regev r = Regev::default_(dev);
        r.randomness().use_discrete_gauss(r.n, r.alpha * r.q);
        auto [rpk, rsk] = r.gen();
        for (int64_t msg : {1, 0, 1}) CHECK(r.dec(rsk, r.enc(rpk, msg)) == msg);
        uint32_t dummy = 0;
        CHECK(qf_sample_discrete_gauss_dev(dev.ctx(), 1, 4, 0, 1, 983, 1, 7.0, &dummy) == QF_ERR_ARG);       // n < 2
        CHECK(qf_sample_discrete_gauss_dev(dev.ctx(), 1, 4, 0, 1, 983, 1024, 1e7, &dummy) == QF_ERR_NOTSUP); // support > 2^21
    });
    run("gaussian_regularity_variants", [&] {  // regev_discrete_gauss.rs / dual_regev_discrete_gauss.rs cycle_* tests
        for (int which = 0; which < 3; which++) {
            RegevWithDiscreteGaussianRegularity r = which == 0 ? RegevWithDiscreteGaussianRegularity::default_(dev)
                                                   : which == 1 ? RegevWithDiscreteGaussianRegularity::new_from_n(dev, 10)
                                                                : RegevWithDiscreteGaussianRegularity::secure128(dev);
            if (which) { r.pp.check_correctness(); r.pp.check_security(); }
            auto [pk, sk] = r.gen();
            CHECK(pk.a.rows == r.pp.n && pk.a.cols == r.pp.m && pk.v.rows == r.pp.m && sk.rows == r.pp.n);
            for (int64_t msg : {0, 1, 1, 0}) CHECK(r.dec(sk, r.enc(pk, msg)) == msg);
            CHECK(r.dec_multiple_bits(sk, r.enc_multiple_bits(pk, 13)) == 13);
            std::vector<uint8_t> msg(100);
            for (size_t i = 0; i < msg.size(); i++) msg[i] = (uint8_t)((i * 7 % 3) == 0);
            const uint64_t pos = r.randomness().take(3, 0);
            DevBuf<uint32_t> C = r.enc_batch(pk, msg);
            CHECK(r.dec_batch(sk, C.get(), msg.size()) == msg);
            // the batch against the oracle on the same SampleD draws (a default-constructed object draws
            // from ChaCha20 under its own fresh key: the oracle restates that generator too)
            std::vector<uint32_t> E(msg.size() * r.pp.m), K((size_t)(r.pp.n + 1) * r.pp.m), ref(msg.size() * (r.pp.n + 1));
            const Seed &sd = r.randomness().seed();
            CHECK(sd.keyed);
            CHECK(qfo_sample_discrete_gauss_keyed(sd.key, sd.seed, 3, pos, E.size(), r.pp.q, r.pp.n, r.pp.r, E.data()) == 0);
            std::copy(pk.a.e.begin(), pk.a.e.end(), K.begin());
            std::copy(pk.v.e.begin(), pk.v.e.end(), K.begin() + (size_t)r.pp.n * r.pp.m);
            qfo_matmul_batch(K.data(), r.pp.n + 1, r.pp.m, r.pp.q, E.data(), 1, (size_t)r.pp.m * 4, msg.size(), nullptr, msg.data(), ref.data());
            CHECK(C.download() == ref);
        }
        for (int which = 0; which < 2; which++) {
            DualRegevWithDiscreteGaussianRegularity d2 = which == 0 ? DualRegevWithDiscreteGaussianRegularity::default_(dev)
                                                                    : DualRegevWithDiscreteGaussianRegularity::new_from_n(dev, 10);
            auto [pk, sk] = d2.gen();
            CHECK(pk.v.rows == d2.pp.n && sk.rows == d2.pp.m);
            for (int64_t msg : {1, 0, 0, 1}) CHECK(d2.dec(sk, d2.enc(pk, msg)) == msg);
        }
        CHECK(throws<Panic>([&] { RegevWithDiscreteGaussianRegularity::new_from_n(dev, 1); }));
        CHECK(throws<MathError>([&] { GaussRegularityParams{2, 16, 443, 4, 0.9}.check_correctness(); }));
    });
    run("classical_gpv_psf", [&] {  // primitive/psf/gpv.rs:189-223
        PSFGPV psf(dev, 4, 64, 60.0);
        CHECK(psf.k == 6 && psf.m_bar == 28 && psf.m() == 52);
        Randomness rng(dev, 5);
        MatZq A(4, psf.m(), 64);
        A.e = rng.uniform(1, (uint64_t)4 * psf.m(), 64);
        MatZ sig; sig.rows = psf.m(); sig.cols = 1; sig.e.resize(psf.m());
        for (uint32_t i = 0; i < psf.m(); i++) sig.e[i] = (int64_t)(i % 7) - 3;
        CHECK(psf.check_domain(sig));
        MatZq t = psf.f_a(A, sig);
        std::vector<uint32_t> x(psf.m()), ref(4);
        for (uint32_t i = 0; i < psf.m(); i++) x[i] = (uint32_t)(((sig.e[i] % 64) + 64) % 64);
        qfo_matvec(A.e.data(), 4, psf.m(), 64, x.data(), ref.data());
        CHECK(t.e == ref);
        MatZ big = sig; big.e[3] = 100000;
        CHECK(!psf.check_domain(big));
        CHECK(throws<Panic>([&] { psf.f_a(A, big); }));
        MatZ shortv = sig; shortv.rows--; shortv.e.pop_back();
        CHECK(!psf.check_domain(shortv));
    });
    run("rot_minus_kat", [&] {  // utils/rotation_matrix.rs:110-121
        std::vector<int64_t> m = rot_minus({1, 5, -1, 9});
        std::vector<int64_t> want = {1, -9, 1, -5, 5, 1, -9, 1, -1, 5, 1, -9, 9, -1, 5, 1};
        CHECK(m == want);
    });

    run("device_key_cache", [&] {  // enc(&pk, ..) twice: the public key is uploaded and recoded once per distinct matrix
        Regev r = Regev::new_from_n(dev, 20);
        auto [pk, sk] = r.gen();
        CHECK(r.dec(sk, r.enc(pk, 1)) == 1);
        qf_key *k1 = dev.cached_key(pk);
        const size_t count = dev.cached_keys();
        for (int64_t msg : {1, 0, 1}) CHECK(r.dec(sk, r.enc(pk, msg)) == msg);
        CHECK(dev.cached_keys() == count);                  // further enc calls upload nothing
        CHECK(dev.cached_key(pk) == k1);                    // hit: same handle
        MatZq pk2 = pk;
        pk2.e[5] = (pk2.e[5] + 1) % pk2.q;                  // one entry differs -> a different key
        CHECK(dev.cached_key(pk2) != k1);
        CHECK(dev.cached_key(pk, true) != k1);              // the transposed upload is its own entry
        std::vector<uint32_t> x = r.randomness().bits(3, r.m), ref(r.n + 1);
        MatZq c = r.enc(pk2, 0, &x);                        // and it computes with ITS contents
        qfo_matvec(pk2.e.data(), r.n + 1, r.m, r.q, x.data(), ref.data());
        CHECK(c.e == ref);
        for (int i = 0; i < 12; i++) {                      // eviction keeps at most 8 resident
            MatZq t = pk; t.e[0] = (uint32_t)i % t.q; (void)dev.cached_key(t);
        }
        CHECK(dev.cached_keys() == 8);
        CHECK(r.dec(sk, r.enc(pk, 1)) == 1);
    });
    run("key_replicate_two_contexts", [&] {  // the multi-GPU step (qf_key_replicate), here two contexts on one device
        Device dev2(0);
        Regev r = Regev::new_from_n(dev, 30);
        auto [pk, sk] = r.gen();
        qf_key *src = dev.cached_key(pk);
        qf_ctx *ctxs[2] = {dev.ctx(), dev2.ctx()};
        qf_key *rep[2] = {nullptr, nullptr};
        CHECK(qf_key_replicate(ctxs, 2, src, rep) == QF_OK);
        std::vector<uint32_t> x = r.randomness().bits(3, r.m), ref(r.n + 1), got(r.n + 1);
        qfo_matvec(pk.e.data(), r.n + 1, r.m, r.q, x.data(), ref.data());
        for (int i = 0; i < 2; i++) {
            uint32_t rows = 0, cols = 0, qq = 0;
            CHECK(rep[i] && qf_key_dims(rep[i], &rows, &cols, &qq) == QF_OK && rows == r.n + 1 && cols == r.m && qq == r.q);
            CHECK(qf_matvec_modq_host(ctxs[i], rep[i], x.data(), got.data()) == QF_OK);
            CHECK(got == ref);
            // and the batched kernel on the replica: B = 40 encryptions against the oracle
            const uint64_t B = 40;
            std::vector<uint8_t> xb(B * bit_stride(r.m)), msg(B), bits;
            qfo_sample_bits(MASTER_SEED, 3, 77, B, r.m, bit_stride(r.m), xb.data());
            for (uint64_t b = 0; b < B; b++) msg[b] = (uint8_t)(b & 1);
            Device &d = i ? dev2 : dev;
            DevBuf<uint8_t> dx(d, xb), dm(d, msg);
            DevBuf<uint32_t> out(d, B * (r.n + 1));
            CHECK(qf_matmul_modq_dev(ctxs[i], rep[i], dx.get(), QF_X_BITS, bit_stride(r.m), 1, B, nullptr, dm.get(), out.get()) == QF_OK);
            std::vector<uint32_t> want(B * (r.n + 1));
            qfo_matmul_batch(pk.e.data(), r.n + 1, r.m, r.q, xb.data(), 0, bit_stride(r.m), B, nullptr, msg.data(), want.data());
            CHECK(out.download() == want);
        }
        qf_key *none[1] = {nullptr};
        CHECK(qf_key_replicate(nullptr, 2, src, rep + 0) == QF_ERR_ARG);
        CHECK(qf_key_replicate(ctxs, 0, src, none) == QF_ERR_ARG);
        for (qf_key *k : rep) qf_key_destroy(k);
    });
    run("default_seeds_are_fresh", [&] {  // two default-constructed schemes must not share secrets (advisor finding)
        Regev a = Regev::default_(dev), b = Regev::default_(dev);
        CHECK(a.randomness().seed().keyed && b.randomness().seed().keyed);   // ChaCha20 under 256 OS bits, not SplitMix64
        CHECK(std::memcmp(a.randomness().seed().key, b.randomness().seed().key, 32) != 0);
        auto ka = a.gen(), kb = b.gen();
        CHECK(ka.second.e != kb.second.e);
        // the public matrix of a default-constructed scheme IS the keyed stream 1 (and nothing SplitMix64 could produce)
        const Seed &sa = a.randomness().seed();
        std::vector<uint32_t> A0((size_t)a.n * a.m);
        qfo_sample_uniform_keyed(sa.key, sa.seed, 1, 0, A0.size(), a.q, A0.data());
        CHECK(std::equal(A0.begin(), A0.end(), ka.first.e.begin()));
        Regev c(dev, 13, 154, 1427, 0.01, 42), d(dev, 13, 154, 1427, 0.01, 42);   // explicit seed: reproducible
        CHECK(c.gen().second.e == d.gen().second.e);
    });

    std::printf("%d checks, %d failures\n", checks, failures);
    return failures ? 1 : 0;
}
