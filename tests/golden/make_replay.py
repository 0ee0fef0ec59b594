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
    open(os.path.join(OUT, "fixtures.txt"), "w").write("\n".join(L) + "\n")
    open(os.path.join(OUT, "replay_test.rs"), "w").write(RUST)


RUST = r'''// Replays tests/replay/fixtures.txt (written by the MI355X build's tests/golden/make_replay.py) through
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
'''

if __name__ == "__main__":
    main()
