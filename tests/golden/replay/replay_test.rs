// Replays tests/replay/fixtures.txt (written by the MI355X build's tests/golden/make_replay.py) through
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
