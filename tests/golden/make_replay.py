"""Writes tests/golden/replay/: small fixtures in qfall-math's own text formats ("[[a, b],[c, d]] mod q",
"[[a],[b]]", "len  c0 c1 ...") plus replay_test.rs, a Rust integration test that feeds them to the REAL
reference.  Nobody could run the Rust side here (no cargo); anyone with a toolchain can:
copy replay/ into the reference's tests/ directory and `cargo test --test replay_test`.
Expected values come from oracle/pyoracle.py (exact big-integer Python); tests/test_oracle_golden.py
checks the C oracle against the files, tests/test_gpu_kernels.py the device.

usage: python tests/golden/make_replay.py        (deterministic; inputs from the seeded generator)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle as po

SEED = 0x5146414C4C
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "replay")


def mat(rows):
    return "[" + ",".join("[" + ", ".join(str(int(v)) for v in r) + "]" for r in rows) + "]"


def poly(c):
    c = [int(v) for v in c]
    while c and c[-1] == 0:
        c.pop()
    return "0" if not c else f"{len(c)}  " + " ".join(str(v) for v in c)


def main():
    os.makedirs(OUT, exist_ok=True)
    L = []
    # Regev::enc  c = pk*x + msg*floor(q/2) on the last entry (regev.rs:396-414)
    n, q = 5, 124991
    m = (n + 1) * 17
    pk = [po.sample_uniform(SEED, 1, i * m, m, q) for i in range(n + 1)]
    x = po.sample_bits(SEED, 3, 0, 1, m)[0]
    c = po.regev_enc(pk, x, 1, q)
    L += ["regev_pk=" + mat(pk) + f" mod {q}", "regev_x=" + mat([[v] for v in x]), "regev_c=" + mat([[v] for v in c]) + f" mod {q}"]
    # SISHash::hash  d = A*v (sis.rs:161-167)
    n, m, q = 4, 40, 257
    A = [po.sample_uniform(SEED, 1, 1000 + i * m, m, q) for i in range(n)]
    v = po.sample_uniform(SEED, 6, 0, m, q)
    d = po.matvec(A, v, q)
    L += ["sis_key=" + mat(A) + f" mod {q}", "sis_value=" + mat([[t] for t in v]) + f" mod {q}", "sis_digest=" + mat([[t] for t in d]) + f" mod {q}"]
    # LPR::enc  c = pk*r + e + msg*floor(q/2) (lpr.rs:394-430), r, e small and signed
    n, q, W = 6, 983, 2
    pk = [po.sample_uniform(SEED, 1, 2000 + i * n, n, q) for i in range(n + 1)]
    r = po.sample_noise(SEED, 3, 0, n, q, W)
    e = po.sample_noise(SEED, 4, 0, n + 1, q, W)
    c = [(a + b) % q for a, b in zip(po.matvec(pk, r, q), e)]
    c[-1] = (c[-1] + q // 2) % q
    L += ["lpr_pk=" + mat(pk) + f" mod {q}", "lpr_r=" + mat([[t] for t in r]) + f" mod {q}", "lpr_e=" + mat([[t] for t in e]) + f" mod {q}",
          "lpr_c=" + mat([[t] for t in c]) + f" mod {q}"]
    # ring product in Z_q[X]/(X^n+1) (ring_lpr.rs:391,455,457): c = a*b
    n, q = 8, 257
    a = po.sample_uniform(SEED, 6, 100, n, q)
    b = po.sample_uniform(SEED, 6, 200, n, q)
    L += [f"ring_n={n}", f"ring_q={q}", "ring_a=" + poly(a), "ring_b=" + poly(b), "ring_c=" + poly(po.polymul_negacyclic(a, b, q))]

    # ---- round 3: every remaining product of SURVEY 8(a) -------------------------------------------
    col = lambda v: mat([[t] for t in v])
    # Regev::dec (regev.rs:438-452) through the scheme's own dec(): five ciphertexts under one secret key whose
    # inner products <[-s;1], c> are 0, floor(q/4), the TIE value (d(r,0) == d(r,floor(q/2)): decides 0), its two
    # neighbours, and a random one.  This is what checks the inferred ring-distance semantics of Zq::distance.
    n, q = 5, 124991
    m = (n + 1) * 17
    s_ = po.sample_uniform(SEED, 2, 0, n, q)
    qh = q // 2
    tie = (q + qh) // 2                       # q - r == r - qh  (q + qh even for this q)
    assert (q + qh) % 2 == 0 and po.ring_dist(tie, 0, q) == po.ring_dist(tie, qh, q)
    targets = [0, q // 4, tie, tie - 1, tie + 1, None]
    L += [f"regev_dec_n={n}", f"regev_dec_m={m}", f"regev_dec_q={q}", "regev_dec_sk=" + col(s_) + f" mod {q}", f"regev_dec_count={len(targets)}"]
    bits = []
    for k, tgt in enumerate(targets):
        c = po.sample_uniform(SEED, 9, 100 * k, n + 1, q)
        if tgt is not None:
            inner = sum((-a) * b for a, b in zip(s_, c[:-1])) % q
            c[-1] = (tgt - inner) % q
            assert (sum((-a) * b for a, b in zip(s_, c[:-1])) + c[-1]) % q == tgt
        bits.append(po.regev_dec(s_, c, q))
        L.append(f"regev_dec_c{k}=" + col(c) + f" mod {q}")
    assert bits[:5] == [0, 0, 0, 1, 0] or bits[2] == 0
    L.append("regev_dec_bits=" + " ".join(str(b) for b in bits))
    # DualRegev::enc / dec (dual_regev.rs:403-410, 437-450): c^t = s^t [A|u] + e^t (+ msg), then c^t [-x;1]
    n, q = 4, 7993
    m = 3 * 13
    A = [po.sample_uniform(SEED, 1, 3000 + i * m, m, q) for i in range(n)]
    xs = po.sample_bits(SEED, 2, 5, 1, m)[0]
    u_ = po.matvec(A, xs, q)
    pk = [A[i] + [u_[i]] for i in range(n)]
    sv = po.sample_uniform(SEED, 3, 50, n, q)
    ev = po.sample_noise(SEED, 4, 50, m + 1, q, 3)
    L += [f"dualregev_n={n}", f"dualregev_m={m}", f"dualregev_q={q}", "dualregev_pk=" + mat(pk) + f" mod {q}", "dualregev_sk=" + col(xs),
          "dualregev_s=" + col(sv) + f" mod {q}", "dualregev_e=" + col(ev) + f" mod {q}"]
    for msg in (0, 1):
        c = po.dual_regev_enc(pk, sv, ev, msg, q)
        assert po.dual_regev_dec(xs, c, q) == msg
        L.append(f"dualregev_c{msg}=" + col(c) + f" mod {q}")
    # RingLPR::enc / dec (ring_lpr.rs:455-457, 489-508) at the default n = 16, q = 2399
    n, q = 16, 2399
    a = po.sample_uniform(SEED, 1, 4000, n, q)
    sk = po.sample_noise(SEED, 2, 4000, n, q, 2)
    b = [(x + y) % q for x, y in zip(po.polymul_negacyclic(a, sk, q), po.sample_noise(SEED, 4, 4000, n, q, 2))]
    r = po.sample_noise(SEED, 3, 4100, n, q, 2)
    e1 = po.sample_noise(SEED, 4, 4200, n, q, 2)
    e2 = po.sample_noise(SEED, 4, 4300, n, q, 2)
    message = 0xB5C3
    mu = po.message_to_mu_bits(message, n)
    u, v = po.ringlpr_enc(a, b, r, e1, e2, mu, q)
    got = sum(bit << i for i, bit in enumerate(po.ringlpr_dec(sk, u, v, q)))
    assert got == message
    L += [f"ringlpr_n={n}", f"ringlpr_q={q}", "ringlpr_a=" + poly(a), "ringlpr_b=" + poly(b), "ringlpr_sk=" + poly(sk), "ringlpr_r=" + poly(r),
          "ringlpr_e1=" + poly(e1), "ringlpr_e2=" + poly(e2), "ringlpr_mu_q_half=" + poly([bit * (q // 2) for bit in mu]),
          f"ringlpr_message={message}", "ringlpr_u=" + poly(u), "ringlpr_v=" + poly(v)]
    # ... and dec on a crafted ciphertext: u = 0, v's coefficients sit on and around both tie values of the rule
    qh = q // 2
    t1 = qh // 2 if qh % 2 == 0 else None                         # r - 0 == qh - r
    t2 = (q + qh) // 2 if (q + qh) % 2 == 0 else None             # q - r == r - qh
    vv = [0, qh, q - 1, 1, qh - 1, qh + 1, q // 4, q // 4 + 1, (3 * q) // 4, (3 * q) // 4 + 1,
          t1 if t1 is not None else 5, t2 if t2 is not None else 6, 600, 1800, 599, 1799]
    bits2 = po.ringlpr_dec(sk, [0] * n, vv, q)
    L += ["ringlpr_tie_v=" + poly(vv), f"ringlpr_tie_message={sum(bit << i for i, bit in enumerate(bits2))}"]
    # PSFGPVRing::f_a (psf/gpv_ring.rs:235-239) at the FDH test's sizes (fdh/gpv_ring.rs:99-123: n = 8, q = 512):
    # a is 1 x (k+2) = 1 x 11 over R_q, sigma an 11 x 1 MatPolyOverZ with signed coefficients
    n, q = 8, 512
    mbar = po.log2_ceil(q) + 2
    Aj = [po.sample_uniform(SEED, 1, 5000 + j * n, n, q) for j in range(mbar)]
    Aj[0] = [1] + [0] * (n - 1)
    sig = [[int(t) - 9 for t in po.sample_uniform(SEED, 7, 5000 + j * n, n, 19)] for j in range(mbar)]
    from fractions import Fraction
    assert po.fdh_check_domain(sig, Fraction(100), mbar)
    t = po.fdh_ring_f_a(Aj, sig, q)
    L += [f"fa_n={n}", f"fa_q={q}", "fa_a=[[" + ", ".join(poly(x) for x in Aj) + "]]", "fa_sigma=[" + ",".join("[" + poly(x) + "]" for x in sig) + "]",
          "fa_t=[[" + poly(t) + "]]"]
    # hash_to_mat_zq_sha256 (hash/sha256.rs:110-133): the FDH shape (n x 1 at q = 512) and a 30-bit modulus
    hm = "replay message 1"
    L += ["hash_msg=" + hm, "hash_8x1_512=" + mat(po.hash_to_mat_zq(hm, 8, 1, 512)) + " mod 512",
          "hash_3x2_big=" + mat(po.hash_to_mat_zq(hm, 3, 2, 1073479681)) + " mod 1073479681"]
    open(os.path.join(OUT, "fixtures.txt"), "w").write("\n".join(L) + "\n")
    open(os.path.join(OUT, "replay_test.rs"), "w").write(RUST)


RUST = r'''// Replays tests/replay/fixtures.txt (written by the MI355X build's tests/golden/make_replay.py) through
// qfall-crypto's own arithmetic.  UNTESTED here (the build container has no Rust toolchain): place this
// file and fixtures.txt under the reference's tests/ and run `cargo test --test replay_test`.
use qfall_crypto::{
    construction::{
        hash::sha256::hash_to_mat_zq_sha256,
        pk_encryption::{DualRegev, PKEncryptionScheme, Regev, RingLPR},
    },
    primitive::psf::{PSFGPVRing, PSF},
    sample::g_trapdoor::gadget_parameters::GadgetParametersRing,
    utils::common_moduli::new_anticyclic,
};
use qfall_math::{
    integer::{MatPolyOverZ, MatZ, PolyOverZ, Z},
    integer_mod_q::{MatPolynomialRingZq, MatZq, ModulusPolynomialRingZq, PolynomialRingZq},
    rational::Q,
    traits::{GetEntry, SetEntry},
};
use std::{collections::HashMap, str::FromStr};

fn fixtures() -> HashMap<String, String> {
    include_str!("fixtures.txt")
        .lines()
        .filter_map(|l| l.split_once('='))
        .map(|(k, v)| (k.to_string(), v.to_string()))
        .collect()
}

#[test]
fn regev_enc_product() {
    let f = fixtures();
    let pk = MatZq::from_str(&f["regev_pk"]).unwrap();
    let x = MatZ::from_str(&f["regev_x"]).unwrap();
    let mut c = &pk * &x; // regev.rs:404
    let q_half = Z::from(&pk.get_mod()).div_floor(&Z::from(2));
    let last: Z = c.get_entry(-1, 0).unwrap();
    c.set_entry(-1, 0, last + q_half).unwrap(); // message bit 1, regev.rs:408-411
    assert_eq!(MatZq::from_str(&f["regev_c"]).unwrap(), c);
}

#[test]
fn sis_hash_product() {
    let f = fixtures();
    let key = MatZq::from_str(&f["sis_key"]).unwrap();
    let value = MatZq::from_str(&f["sis_value"]).unwrap();
    assert_eq!(MatZq::from_str(&f["sis_digest"]).unwrap(), &key * &value); // sis.rs:166
}

#[test]
fn lpr_enc_product() {
    let f = fixtures();
    let pk = MatZq::from_str(&f["lpr_pk"]).unwrap();
    let r = MatZq::from_str(&f["lpr_r"]).unwrap();
    let e = MatZq::from_str(&f["lpr_e"]).unwrap();
    let mut c = &pk * &r + &e; // lpr.rs:420
    let q_half = Z::from(&pk.get_mod()).div_floor(&Z::from(2));
    let last: Z = c.get_entry(-1, 0).unwrap();
    c.set_entry(-1, 0, last + q_half).unwrap();
    assert_eq!(MatZq::from_str(&f["lpr_c"]).unwrap(), c);
}

#[test]
fn ring_product() {
    let f = fixtures();
    let n: i64 = f["ring_n"].parse().unwrap();
    let q: i64 = f["ring_q"].parse().unwrap();
    let modulus = new_anticyclic(n, q).unwrap(); // common_moduli.rs:41-48
    let a = PolynomialRingZq::from((&PolyOverZ::from_str(&f["ring_a"]).unwrap(), &modulus));
    let b = PolynomialRingZq::from((&PolyOverZ::from_str(&f["ring_b"]).unwrap(), &modulus));
    let c = PolynomialRingZq::from((&PolyOverZ::from_str(&f["ring_c"]).unwrap(), &modulus));
    assert_eq!(c, &a * &b); // ring_lpr.rs:391
}

// ---- round 3: the remaining products of the hot path, through the schemes' own dec() where it is deterministic

#[test]
fn regev_dec_decision_rule() {
    // regev.rs:438-452; ciphertext 2 is the TIE of the rule (distance(0) == distance(q/2)): must decide 0.
    // This is the check of Zq::distance's wrap-around semantics the MI355X build could only infer.
    let f = fixtures();
    let (n, m, q): (i64, i64, i64) = (f["regev_dec_n"].parse().unwrap(), f["regev_dec_m"].parse().unwrap(), f["regev_dec_q"].parse().unwrap());
    let regev = Regev::new(n, m, q, Q::from(0.01_f64));
    let sk = MatZq::from_str(&f["regev_dec_sk"]).unwrap();
    let want: Vec<i64> = f["regev_dec_bits"].split(' ').map(|b| b.parse().unwrap()).collect();
    for (k, bit) in want.iter().enumerate() {
        let c = MatZq::from_str(&f[&format!("regev_dec_c{k}")]).unwrap();
        assert_eq!(Z::from(*bit), regev.dec(&sk, &c), "ciphertext {k}");
    }
}

#[test]
fn dual_regev_enc_product_and_dec() {
    let f = fixtures();
    let (n, m, q): (i64, i64, i64) = (f["dualregev_n"].parse().unwrap(), f["dualregev_m"].parse().unwrap(), f["dualregev_q"].parse().unwrap());
    let scheme = DualRegev::new(n, m, q, Q::from(0.01_f64));
    let pk = MatZq::from_str(&f["dualregev_pk"]).unwrap();
    let sk = MatZ::from_str(&f["dualregev_sk"]).unwrap();
    let s = MatZq::from_str(&f["dualregev_s"]).unwrap();
    let e = MatZq::from_str(&f["dualregev_e"]).unwrap();
    let q_half = Z::from(&pk.get_mod()).div_floor(&Z::from(2));
    for msg in 0..2 {
        let mut c = (s.transpose() * &pk + e.transpose()).transpose(); // dual_regev.rs:403
        let last: Z = c.get_entry(-1, 0).unwrap();
        c.set_entry(-1, 0, last + Z::from(msg) * &q_half).unwrap(); // :406-410
        assert_eq!(MatZq::from_str(&f[&format!("dualregev_c{msg}")]).unwrap(), c);
        assert_eq!(Z::from(msg), scheme.dec(&sk, &c)); // :437-450
    }
}

fn ring_elem(f: &HashMap<String, String>, key: &str, modulus: &ModulusPolynomialRingZq) -> PolynomialRingZq {
    PolynomialRingZq::from((&PolyOverZ::from_str(&f[key]).unwrap(), modulus))
}

#[test]
fn ring_lpr_enc_products_and_dec() {
    let f = fixtures();
    let (n, q): (i64, i64) = (f["ringlpr_n"].parse().unwrap(), f["ringlpr_q"].parse().unwrap());
    let scheme = RingLPR::new(n, q, Q::from(0.0039_f64));
    let modulus = new_anticyclic(n, q).unwrap();
    let a = ring_elem(&f, "ringlpr_a", &modulus);
    let b = ring_elem(&f, "ringlpr_b", &modulus);
    let sk = ring_elem(&f, "ringlpr_sk", &modulus);
    let r = ring_elem(&f, "ringlpr_r", &modulus);
    let e1 = ring_elem(&f, "ringlpr_e1", &modulus);
    let e2 = ring_elem(&f, "ringlpr_e2", &modulus);
    let mu_q_half = ring_elem(&f, "ringlpr_mu_q_half", &modulus);
    let u = &a * &r + e1; // ring_lpr.rs:455
    let v = &b * &r + e2 + mu_q_half; // :457
    assert_eq!(ring_elem(&f, "ringlpr_u", &modulus), u);
    assert_eq!(ring_elem(&f, "ringlpr_v", &modulus), v);
    let message: i64 = f["ringlpr_message"].parse().unwrap();
    assert_eq!(Z::from(message), scheme.dec(&sk, &(u, v))); // :489-508
    // coefficients on and around the ties of the per-coefficient rule (u = 0, so v - s*u = v)
    let zero = PolynomialRingZq::from((&PolyOverZ::from_str("0").unwrap(), &modulus));
    let tie_message: i64 = f["ringlpr_tie_message"].parse().unwrap();
    assert_eq!(Z::from(tie_message), scheme.dec(&sk, &(zero, ring_elem(&f, "ringlpr_tie_v", &modulus))));
}

#[test]
fn psf_gpv_ring_f_a() {
    // primitive/psf/gpv_ring.rs:235-239 at the sizes of signature/fdh/gpv_ring.rs:99-123
    let f = fixtures();
    let (n, q): (i64, i64) = (f["fa_n"].parse().unwrap(), f["fa_q"].parse().unwrap());
    let psf = PSFGPVRing { gp: GadgetParametersRing::init_default(n, q), s: Q::from(100), s_td: Q::from(1.005_f64) };
    let a = MatPolynomialRingZq::from((&MatPolyOverZ::from_str(&f["fa_a"]).unwrap(), &psf.gp.modulus));
    let sigma = MatPolyOverZ::from_str(&f["fa_sigma"]).unwrap();
    assert!(psf.check_domain(&sigma));
    let t = MatPolynomialRingZq::from((&MatPolyOverZ::from_str(&f["fa_t"]).unwrap(), &psf.gp.modulus));
    assert_eq!(t, psf.f_a(&a, &sigma));
}

#[test]
fn hash_to_mat_zq() {
    let f = fixtures(); // hash/sha256.rs:110-133
    assert_eq!(MatZq::from_str(&f["hash_8x1_512"]).unwrap(), hash_to_mat_zq_sha256(&f["hash_msg"], 8, 1, 512));
    assert_eq!(MatZq::from_str(&f["hash_3x2_big"]).unwrap(), hash_to_mat_zq_sha256(&f["hash_msg"], 3, 2, 1073479681));
}
'''

if __name__ == "__main__":
    main()
