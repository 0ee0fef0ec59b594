// Replays tests/replay/fixtures.txt (written by the MI355X build's tests/golden/make_replay.py) through
// qfall-crypto's own arithmetic.  UNTESTED here (the build container has no Rust toolchain): place this
// file and fixtures.txt under the reference's tests/ and run `cargo test --test replay_test`.
use qfall_crypto::utils::common_moduli::new_anticyclic;
use qfall_math::{
    integer::{MatZ, PolyOverZ, Z},
    integer_mod_q::{MatZq, PolynomialRingZq},
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
