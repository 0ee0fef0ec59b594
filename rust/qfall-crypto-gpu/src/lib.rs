//! qFALL-crypto's hot path on an MI355X, behind the reference's own traits.
//!
//! UNCOMPILED in this repository (no Rust toolchain in the build image).  Every method below is a
//! transliteration of a method of the tested C++ host layer (`include/qfall_crypto.hpp`, run on the GPU by
//! `tests/cpp/test_host.cpp`) and calls the C ABI exactly as that layer does.
//!
//! What stays upstream: parameter generation, every sampler (the reference draws from the OS CSPRNG inside
//! qfall-math: regev.rs:352-364,401; those calls are kept verbatim, so the randomness of a GPU-backed scheme
//! IS the reference's), trapdoor generation and signing.  What moves to the device: the products --
//! `pk * x`, `pk * r + e`, `s^t * pk + e^t`, `[-s;1] . c`, `a*r + e1`, `b*r + e2 + mu`, `v - s*u`, `a * sigma`,
//! `key * value` -- single messages through the `_host` entry points, batches through the `_dev` ones.
//!
//! Error behaviour: the reference panics on dimension / modulus mismatch (hash/sis.rs:157-165); a nonzero
//! return code of the library becomes the same kind of panic, with the library's message.
use qfall_crypto::construction::{
    hash::{sha256::HashMatPolynomialRingZq, HashInto},
    pk_encryption::{DualRegev, PKEncryptionScheme, Regev, RingLPR, LPR},
    signature::{SignatureScheme, FDH},
};
use qfall_crypto::primitive::psf::{PSFGPVRing, PSF};
use qfall_math::{
    integer::{MatPolyOverZ, MatZ, PolyOverZ, Z},
    integer_mod_q::{MatPolynomialRingZq, MatZq, Modulus, ModulusPolynomialRingZq, PolynomialRingZq, Zq},
    rational::Q,
    traits::{Concatenate, GetCoefficient, GetEntry, GetNumColumns, GetNumRows, Pow, SetCoefficient, SetEntry},
};
use qfgpu_sys::*;
use std::{cell::RefCell, ffi::CStr, os::raw::c_void, ptr, rc::Rc};

// ------------------------------------------------------------------------------------------------
// device context and buffers
// ------------------------------------------------------------------------------------------------

/// One GPU, one stream (`qf_ctx`).  Not `Send`: like the reference's scheme structs (Rc inside `Modulus`).
pub struct Device {
    ctx: *mut qf_ctx,
}

impl Device {
    pub fn new(device: i32) -> Rc<Self> {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { qf_ctx_create(device, ptr::null_mut(), &mut ctx) };
        assert!(rc == QF_OK, "qf_ctx_create failed with code {rc}");
        Rc::new(Self { ctx })
    }
    /// nonzero code -> panic with the library's text (the reference panics in the same situations)
    fn check(&self, rc: i32) {
        if rc != QF_OK {
            let msg = unsafe { CStr::from_ptr(qf_last_error(self.ctx)) }.to_string_lossy().into_owned();
            panic!("libqfgpu error {rc}: {msg}");
        }
    }
    fn upload<T: Copy>(&self, host: &[T]) -> DevBuf<T> {
        let buf = DevBuf::<T>::new(self, host.len());
        self.check(unsafe { qf_memcpy_h2d(self.ctx, buf.ptr as *mut c_void, host.as_ptr() as *const c_void, std::mem::size_of_val(host)) });
        buf
    }
}
impl Drop for Device {
    fn drop(&mut self) {
        unsafe { qf_ctx_destroy(self.ctx) }
    }
}

/// Device memory owned through `qf_malloc` / `qf_free`.
pub struct DevBuf<T> {
    ctx: *mut qf_ctx,
    ptr: *mut T,
    len: usize,
}
impl<T: Copy> DevBuf<T> {
    fn new(dev: &Device, len: usize) -> Self {
        let mut p: *mut c_void = ptr::null_mut();
        dev.check(unsafe { qf_malloc(dev.ctx, len.max(1) * std::mem::size_of::<T>(), &mut p) });
        Self { ctx: dev.ctx, ptr: p as *mut T, len }
    }
    fn download(&self, dev: &Device) -> Vec<T> {
        let mut v = Vec::<T>::with_capacity(self.len);
        dev.check(unsafe { qf_memcpy_d2h(self.ctx, v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * std::mem::size_of::<T>()) });
        unsafe { v.set_len(self.len) };
        v
    }
}
impl<T> Drop for DevBuf<T> {
    fn drop(&mut self) {
        unsafe { qf_free(self.ctx, self.ptr as *mut c_void) };
    }
}

// ------------------------------------------------------------------------------------------------
// conversions at the seam: MatZq / MatZ / PolynomialRingZq / MatPolyOverZ <-> flat words (O(size))
// ------------------------------------------------------------------------------------------------

fn modulus_u32(q: &Modulus) -> u32 {
    let q = u32::try_from(&Z::from(q)).expect("the GPU path needs q < 2^31"); // QF_ERR_MODULUS otherwise
    assert!(q < (1 << 31), "the GPU path needs q < 2^31");
    q
}
/// row-major canonical residues
fn matzq_to_u32(m: &MatZq) -> (Vec<u32>, u32, u32, u32) {
    let (rows, cols) = (m.get_num_rows(), m.get_num_columns());
    let mut v = Vec::with_capacity((rows * cols) as usize);
    for i in 0..rows {
        for j in 0..cols {
            let e: Z = m.get_entry(i, j).unwrap();
            v.push(u32::try_from(&e).unwrap());
        }
    }
    (v, rows as u32, cols as u32, modulus_u32(&m.get_mod()))
}
fn u32_to_matzq(v: &[u32], rows: i64, cols: i64, q: &Modulus) -> MatZq {
    let mut m = MatZq::new(rows, cols, q);
    for i in 0..rows {
        for j in 0..cols {
            m.set_entry(i, j, Z::from(v[(i * cols + j) as usize])).unwrap();
        }
    }
    m
}
/// {0,1} column vector of MatZ -> words (regev.rs:401's x, dual_regev's sk)
fn matz_col_to_u32(m: &MatZ, q: u32) -> Vec<u32> {
    (0..m.get_num_rows())
        .map(|i| {
            let e: Z = m.get_entry(i, 0).unwrap();
            u32::try_from(&e.modulo(q)).unwrap()
        })
        .collect()
}
/// n coefficients, degree-ascending, canonical
fn poly_to_u32(p: &PolynomialRingZq, n: i64) -> Vec<u32> {
    (0..n)
        .map(|i| {
            let c: Z = p.get_coeff(i).unwrap();
            u32::try_from(&c).unwrap()
        })
        .collect()
}
fn u32_to_poly(v: &[u32], modulus: &ModulusPolynomialRingZq) -> PolynomialRingZq {
    let mut p = PolyOverZ::default();
    for (i, c) in v.iter().enumerate() {
        p.set_coeff(i as i64, Z::from(*c)).unwrap();
    }
    PolynomialRingZq::from((&p, modulus))
}

/// A `MatZq` key resident on the device, re-uploaded only when the matrix changes (the C++ layer keeps a
/// content-verified LRU of 8; one slot is enough to show the shape).
struct KeyCache {
    slot: RefCell<Option<(MatZq, bool, *mut qf_key)>>,
}
impl KeyCache {
    fn new() -> Self {
        Self { slot: RefCell::new(None) }
    }
    fn get(&self, dev: &Device, m: &MatZq, transpose: bool) -> *mut qf_key {
        if let Some((cached, t, key)) = self.slot.borrow().as_ref() {
            if *t == transpose && cached == m {
                return *key;
            }
        }
        let (words, rows, cols, q) = matzq_to_u32(m);
        let mut key = ptr::null_mut();
        dev.check(unsafe { qf_key_upload_host(dev.ctx, words.as_ptr(), rows, cols, q, transpose as i32, &mut key) });
        if let Some((_, _, old)) = self.slot.replace(Some((m.clone(), transpose, key))) {
            unsafe { qf_key_destroy(old) };
        }
        key
    }
}
impl Drop for KeyCache {
    fn drop(&mut self) {
        if let Some((_, _, key)) = self.slot.borrow_mut().take() {
            unsafe { qf_key_destroy(key) };
        }
    }
}

/// `[dist(r,0) > dist(r, floor(q/2))]` for ONE ciphertext through the batched kernel (regev.rs:438-452):
/// w = -sk mod q, C = the ciphertext as one row of n+1 words.
fn decode_one(dev: &Device, w: &[u32], c: &[u32], q: u32) -> Z {
    let (dw, dc) = (dev.upload(w), dev.upload(c));
    let bits = DevBuf::<u8>::new(dev, 1);
    dev.check(unsafe { qf_dot_decode_dev(dev.ctx, dw.ptr, w.len() as u32, q, dc.ptr, 1, bits.ptr) });
    Z::from(bits.download(dev)[0])
}
fn add_message(c: &mut [u32], message: &Z, q: u32) {
    // hide the message in the last entry: + msg * floor(q/2)  (regev.rs:408-411)
    if message == &Z::ONE {
        let last = c.len() - 1;
        c[last] = ((c[last] as u64 + (q / 2) as u64) % q as u64) as u32;
    }
}

// ------------------------------------------------------------------------------------------------
// Regev (construction/pk_encryption/regev.rs)
// ------------------------------------------------------------------------------------------------

/// The reference's scheme structs keep their parameters private (regev.rs:50-55), so the GPU types hold their
/// own copies next to an `inner` instance built from the same values (for `check_correctness`, `check_security`
/// and everything else that is not on the hot path).
pub struct RegevGpu {
    pub inner: Regev,
    n: Z,
    m: Z,
    q: Modulus,
    alpha: Q,
    dev: Rc<Device>,
    keys: KeyCache,
}
impl RegevGpu {
    /// Regev::new(n, m, q, alpha)  (regev.rs:87-99)
    pub fn new(dev: Rc<Device>, n: impl Into<Z>, m: impl Into<Z>, q: impl Into<Modulus>, alpha: impl Into<Q>) -> Self {
        let (n, m, q, alpha): (Z, Z, Modulus, Q) = (n.into(), m.into(), q.into(), alpha.into());
        Self { inner: Regev::new(n.clone(), m.clone(), q.clone(), alpha.clone()), n, m, q, alpha, dev, keys: KeyCache::new() }
    }
    /// B one-bit messages in ONE launch of kernel (a) (`qf_matmul_modq_dev`, bit operands): x_bits_dev holds B
    /// rows of `stride` bytes (a multiple of 16) with the bits of each message's x, out_dev receives B rows of
    /// n+1 words.  Buffers are the caller's device memory.
    pub fn enc_batch(&self, pk: &MatZq, x_bits_dev: *const c_void, stride: usize, msg_bits_dev: *const u8, b: u64, out_dev: *mut u32) {
        let key = self.keys.get(&self.dev, pk, false);
        self.dev.check(unsafe { qf_matmul_modq_dev(self.dev.ctx, key, x_bits_dev, QF_X_BITS, stride, 0, b, ptr::null(), msg_bits_dev, out_dev) });
    }
    /// The same with x drawn on the device as Regev::enc does (regev.rs:401), under `key32` (32 bytes of the OS CSPRNG)
    /// and `nonce`: the sampler writes the bits k-tile-major (`QF_X_BITS_KT`, the form kernel (a) reads fastest) into
    /// `planes_dev` = ceil(m/128) planes of 16*b bytes, the caller's device memory like the other buffers.
    pub fn enc_batch_sampled(&self, pk: &MatZq, key32: &[u8; 32], nonce: u64, first_row: u64, planes_dev: *mut u8, msg_bits_dev: *const u8, b: u64, out_dev: *mut u32) {
        let key = self.keys.get(&self.dev, pk, false);
        let m = u32::try_from(&self.m).expect("m fits u32");
        let kt_stride = 16 * b as usize;
        self.dev.check(unsafe { qf_sample_bits_kt_keyed_dev(self.dev.ctx, key32.as_ptr(), nonce, 3, first_row, b, m, kt_stride, planes_dev) });
        self.dev.check(unsafe { qf_matmul_modq_dev(self.dev.ctx, key, planes_dev as *const c_void, QF_X_BITS_KT, kt_stride, 0, b, ptr::null(), msg_bits_dev, out_dev) });
    }
    /// B ciphertexts (rows of n+1 words in device memory) -> B bytes 0/1
    pub fn dec_batch(&self, sk: &MatZq, c_dev: *const u32, b: u64, bits_dev: *mut u8) {
        let (s, n, _, q) = matzq_to_u32(sk);
        let w: Vec<u32> = s.iter().map(|v| (q - v) % q).collect();
        let dw = self.dev.upload(&w);
        self.dev.check(unsafe { qf_dot_decode_dev(self.dev.ctx, dw.ptr, n, q, c_dev, b, bits_dev) });
        self.dev.check(unsafe { qf_sync(self.dev.ctx) });
    }
}
impl PKEncryptionScheme for RegevGpu {
    type PublicKey = MatZq;
    type SecretKey = MatZq;
    type Cipher = MatZq;

    /// regev.rs:350-374: A, s, e are sampled upstream; b^t = s^t*A + e^t is the library's x^t*A product
    fn gen(&self) -> (Self::PublicKey, Self::SecretKey) {
        let (n, m, q, alpha) = (&self.n, &self.m, &self.q, &self.alpha);
        let mat_a = MatZq::sample_uniform(n, m, q);
        let vec_s = MatZq::sample_uniform(n, 1, q);
        let vec_e_t = MatZq::sample_discrete_gauss(1, m, q, n, 0, alpha * Z::from(q)).unwrap();
        let qq = modulus_u32(q);
        let key = self.keys.get(&self.dev, &mat_a, false);
        let (s_words, ..) = matzq_to_u32(&vec_s);
        let mut b_t = vec![0u32; i64::try_from(m).unwrap() as usize];
        self.dev.check(unsafe { qf_vecmat_modq_host(self.dev.ctx, key, s_words.as_ptr(), b_t.as_mut_ptr()) }); // regev.rs:367
        let (e_words, ..) = matzq_to_u32(&vec_e_t);
        for (b, e) in b_t.iter_mut().zip(e_words.iter()) {
            *b = ((*b as u64 + *e as u64) % qq as u64) as u32;
        }
        let vec_b_t = u32_to_matzq(&b_t, 1, i64::try_from(m).unwrap(), q);
        (mat_a.concat_vertical(&vec_b_t).unwrap(), vec_s) // regev.rs:370-373
    }

    fn enc(&self, pk: &Self::PublicKey, message: impl Into<Z>) -> Self::Cipher {
        let message: Z = message.into().modulo(2); // regev.rs:398
        let vec_x = MatZ::sample_uniform(&self.m, 1, 0, 2).unwrap(); // regev.rs:401 (the reference's own draw)
        let q = modulus_u32(&self.q);
        let key = self.keys.get(&self.dev, pk, false);
        let x = matz_col_to_u32(&vec_x, q);
        let mut c = vec![0u32; pk.get_num_rows() as usize];
        self.dev.check(unsafe { qf_matvec_modq_host(self.dev.ctx, key, x.as_ptr(), c.as_mut_ptr()) }); // regev.rs:404
        add_message(&mut c, &message, q);
        u32_to_matzq(&c, pk.get_num_rows(), 1, &self.q)
    }

    fn dec(&self, sk: &Self::SecretKey, cipher: &Self::Cipher) -> Z {
        let (s, _, _, q) = matzq_to_u32(sk);
        let (c, ..) = matzq_to_u32(cipher);
        assert!(c.len() == s.len() + 1, "mismatching dimensions of secret key and cipher"); // dot_product panics upstream
        let w: Vec<u32> = s.iter().map(|v| (q - v) % q).collect(); // Z::MINUS_ONE * sk, regev.rs:439
        decode_one(&self.dev, &w, &c, q)
    }
}

// ------------------------------------------------------------------------------------------------
// LPR (construction/pk_encryption/lpr.rs): pk (n+1) x n, r and e from chi
// ------------------------------------------------------------------------------------------------

pub struct LPRGpu {
    pub inner: LPR,
    n: Z,
    q: Modulus,
    alpha: Q,
    dev: Rc<Device>,
    keys: KeyCache,
}
impl LPRGpu {
    /// LPR::new(n, q, alpha)
    pub fn new(dev: Rc<Device>, n: impl Into<Z>, q: impl Into<Modulus>, alpha: impl Into<Q>) -> Self {
        let (n, q, alpha): (Z, Modulus, Q) = (n.into(), q.into(), alpha.into());
        Self { inner: LPR::new(n.clone(), q.clone(), alpha.clone()), n, q, alpha, dev, keys: KeyCache::new() }
    }
    /// B encryptions in one launch: r_dev B x n words, e_dev B x (n+1) words, x_digits = 2 promises |r| < 2^15
    /// (verified on the device; a broken promise is redone at full width).  lpr.rs:420-427.
    pub fn enc_batch(&self, pk: &MatZq, r_dev: *const u32, e_dev: *const u32, msg_bits_dev: *const u8, b: u64, out_dev: *mut u32) {
        let key = self.keys.get(&self.dev, pk, false);
        let stride = pk.get_num_columns() as usize * 4;
        self.dev.check(unsafe { qf_matmul_modq_dev(self.dev.ctx, key, r_dev as *const c_void, QF_X_U32, stride, 2, b, e_dev, msg_bits_dev, out_dev) });
    }
}
impl PKEncryptionScheme for LPRGpu {
    type PublicKey = MatZq;
    type SecretKey = MatZq;
    type Cipher = MatZq;

    fn gen(&self) -> (Self::PublicKey, Self::SecretKey) {
        // lpr.rs:338-368: as Regev::gen with m = n and s drawn from chi
        let (n, q, alpha) = (&self.n, &self.q, &self.alpha);
        let mat_a = MatZq::sample_uniform(n, n, q);
        let vec_s = MatZq::sample_discrete_gauss(n, 1, q, n, 0, alpha * Z::from(q)).unwrap();
        let vec_e_t = MatZq::sample_discrete_gauss(1, n, q, n, 0, alpha * Z::from(q)).unwrap();
        let qq = modulus_u32(q);
        let key = self.keys.get(&self.dev, &mat_a, false);
        let (s_words, ..) = matzq_to_u32(&vec_s);
        let cols = i64::try_from(n).unwrap();
        let mut b_t = vec![0u32; cols as usize];
        self.dev.check(unsafe { qf_vecmat_modq_host(self.dev.ctx, key, s_words.as_ptr(), b_t.as_mut_ptr()) }); // lpr.rs:363
        let (e_words, ..) = matzq_to_u32(&vec_e_t);
        for (b, e) in b_t.iter_mut().zip(e_words.iter()) {
            *b = ((*b as u64 + *e as u64) % qq as u64) as u32;
        }
        (mat_a.concat_vertical(&u32_to_matzq(&b_t, 1, cols, q)).unwrap(), vec_s)
    }

    fn enc(&self, pk: &Self::PublicKey, message: impl Into<Z>) -> Self::Cipher {
        let message: Z = message.into().modulo(2);
        let (n, q, alpha) = (&self.n, &self.q, &self.alpha);
        let vec_r = MatZq::sample_discrete_gauss(n, 1, q, n, 0, alpha * Z::from(q)).unwrap(); // lpr.rs:399-407
        let vec_e = MatZq::sample_discrete_gauss(n + 1, 1, q, n, 0, alpha * Z::from(q)).unwrap(); // lpr.rs:409-417
        let qq = modulus_u32(q);
        let key = self.keys.get(&self.dev, pk, false);
        let (r, ..) = matzq_to_u32(&vec_r);
        let (e, ..) = matzq_to_u32(&vec_e);
        let mut c = vec![0u32; pk.get_num_rows() as usize];
        self.dev.check(unsafe { qf_matvec_modq_host(self.dev.ctx, key, r.as_ptr(), c.as_mut_ptr()) }); // pk * vec_r, lpr.rs:420
        for (ci, ei) in c.iter_mut().zip(e.iter()) {
            *ci = ((*ci as u64 + *ei as u64) % qq as u64) as u32; // + vec_e
        }
        add_message(&mut c, &message, qq); // lpr.rs:424-427
        u32_to_matzq(&c, pk.get_num_rows(), 1, q)
    }

    fn dec(&self, sk: &Self::SecretKey, cipher: &Self::Cipher) -> Z {
        let (s, _, _, q) = matzq_to_u32(sk); // lpr.rs:454-468: same rule as Regev
        let (c, ..) = matzq_to_u32(cipher);
        assert!(c.len() == s.len() + 1, "mismatching dimensions of secret key and cipher");
        let w: Vec<u32> = s.iter().map(|v| (q - v) % q).collect();
        decode_one(&self.dev, &w, &c, q)
    }
}

// ------------------------------------------------------------------------------------------------
// DualRegev (construction/pk_encryption/dual_regev.rs): pk = [A | u] n x (m+1), sk = x in {0,1}^m
// ------------------------------------------------------------------------------------------------

pub struct DualRegevGpu {
    pub inner: DualRegev,
    n: Z,
    m: Z,
    q: Modulus,
    alpha: Q,
    dev: Rc<Device>,
    keys: KeyCache,   // pk as stored (u = A*x in gen)
    keys_t: KeyCache, // pk transposed: the batched form of enc
}
impl DualRegevGpu {
    /// DualRegev::new(n, m, q, alpha)  (dual_regev.rs:87-99)
    pub fn new(dev: Rc<Device>, n: impl Into<Z>, m: impl Into<Z>, q: impl Into<Modulus>, alpha: impl Into<Q>) -> Self {
        let (n, m, q, alpha): (Z, Z, Modulus, Q) = (n.into(), m.into(), q.into(), alpha.into());
        Self { inner: DualRegev::new(n.clone(), m.clone(), q.clone(), alpha.clone()), n, m, q, alpha, dev, keys: KeyCache::new(), keys_t: KeyCache::new() }
    }
    /// B encryptions in one launch: the key is the TRANSPOSED pk ((m+1) x n), s_dev B x n words (uniform in
    /// Z_q: full width), e_dev B x (m+1) words; dual_regev.rs:403-410.  At secure128 this runs the short-K
    /// sweep kernel (key tile resident in LDS).
    pub fn enc_batch(&self, pk: &MatZq, s_dev: *const u32, e_dev: *const u32, msg_bits_dev: *const u8, b: u64, out_dev: *mut u32) {
        let key = self.keys_t.get(&self.dev, pk, true);
        let stride = pk.get_num_rows() as usize * 4;
        self.dev.check(unsafe { qf_matmul_modq_dev(self.dev.ctx, key, s_dev as *const c_void, QF_X_U32, stride, 0, b, e_dev, msg_bits_dev, out_dev) });
    }
}
impl PKEncryptionScheme for DualRegevGpu {
    type PublicKey = MatZq;
    type SecretKey = MatZ;
    type Cipher = MatZq;

    fn gen(&self) -> (Self::PublicKey, Self::SecretKey) {
        // dual_regev.rs:348-358: A uniform, x <- {0,1}^m, u = A*x
        let (n, m, q) = (&self.n, &self.m, &self.q);
        let mat_a = MatZq::sample_uniform(n, m, q);
        let vec_x = MatZ::sample_uniform(m, 1, 0, 2).unwrap();
        let qq = modulus_u32(q);
        let key = self.keys.get(&self.dev, &mat_a, false);
        let x = matz_col_to_u32(&vec_x, qq);
        let rows = i64::try_from(n).unwrap();
        let mut u = vec![0u32; rows as usize];
        self.dev.check(unsafe { qf_matvec_modq_host(self.dev.ctx, key, x.as_ptr(), u.as_mut_ptr()) }); // dual_regev.rs:354
        (mat_a.concat_horizontal(&u32_to_matzq(&u, rows, 1, q)).unwrap(), vec_x)
    }

    fn enc(&self, pk: &Self::PublicKey, message: impl Into<Z>) -> Self::Cipher {
        let message: Z = message.into().modulo(2);
        let (n, m, q, alpha) = (&self.n, &self.m, &self.q, &self.alpha);
        let vec_s_t = MatZq::sample_uniform(1, n, q); // dual_regev.rs:389
        let vec_e_t = MatZq::sample_discrete_gauss(1, m + 1, q, n, 0, alpha * Z::from(q)).unwrap(); // :391-399
        let qq = modulus_u32(q);
        let key = self.keys.get(&self.dev, pk, false);
        let (s, ..) = matzq_to_u32(&vec_s_t);
        let (e, ..) = matzq_to_u32(&vec_e_t);
        let mut c = vec![0u32; pk.get_num_columns() as usize];
        self.dev.check(unsafe { qf_vecmat_modq_host(self.dev.ctx, key, s.as_ptr(), c.as_mut_ptr()) }); // vec_s_t * pk, :403
        for (ci, ei) in c.iter_mut().zip(e.iter()) {
            *ci = ((*ci as u64 + *ei as u64) % qq as u64) as u32;
        }
        add_message(&mut c, &message, qq); // :406-410
        u32_to_matzq(&c, pk.get_num_columns(), 1, q)
    }

    fn dec(&self, sk: &Self::SecretKey, cipher: &Self::Cipher) -> Z {
        let (c, _, _, q) = matzq_to_u32(cipher); // dual_regev.rs:437-450: c^t * [-x ; 1]
        let x = matz_col_to_u32(sk, q);
        assert!(c.len() == x.len() + 1, "mismatching dimensions of secret key and cipher");
        let w: Vec<u32> = x.iter().map(|v| (q - v) % q).collect();
        decode_one(&self.dev, &w, &c, q)
    }
}

// ------------------------------------------------------------------------------------------------
// RingLPR (construction/pk_encryption/ring_lpr.rs) over R_q = Z_q[X]/(X^n+1)
// ------------------------------------------------------------------------------------------------

pub struct RingLPRGpu {
    pub inner: RingLPR,
    nz: Z,
    modulus: ModulusPolynomialRingZq,
    alpha: Q,
    dev: Rc<Device>,
    ring: *mut qf_ring,
    n: i64,
    q: u32,
}
impl RingLPRGpu {
    /// RingLPR::new(n, q, alpha)  (ring_lpr.rs:89-98)
    pub fn new(dev: Rc<Device>, n: impl Into<Z>, q: impl Into<Modulus>, alpha: impl Into<Q>) -> Self {
        let (nz, qm, alpha): (Z, Modulus, Q) = (n.into(), q.into(), alpha.into());
        let modulus = qfall_crypto::utils::common_moduli::new_anticyclic(&nz, &qm).unwrap(); // (X^n + 1) mod q, as upstream
        let n = i64::try_from(&nz).unwrap();
        let q = modulus_u32(&qm);
        let mut ring = ptr::null_mut();
        dev.check(unsafe { qf_ring_create(dev.ctx, n as u32, q, &mut ring) }); // the device-side new_anticyclic, common_moduli.rs:41-48
        Self { inner: RingLPR::new(nz.clone(), qm, alpha.clone()), nz, modulus, alpha, dev, ring, n, q }
    }
    fn rkey(&self, polys: &[&PolynomialRingZq]) -> *mut qf_rkey {
        let words: Vec<u32> = polys.iter().flat_map(|p| poly_to_u32(p, self.n)).collect();
        let mut k = ptr::null_mut();
        self.dev.check(unsafe { qf_rkey_upload_host(self.dev.ctx, self.ring, words.as_ptr(), polys.len() as u32, &mut k) });
        k
    }
    /// one product a*b in R_q on the device (ring_lpr.rs:391)
    fn mul(&self, a: &PolynomialRingZq, b: &PolynomialRingZq) -> Vec<u32> {
        let (da, db) = (self.dev.upload(&poly_to_u32(a, self.n)), self.dev.upload(&poly_to_u32(b, self.n)));
        let dc = DevBuf::<u32>::new(&self.dev, self.n as usize);
        self.dev.check(unsafe { qf_polymul_dev(self.dev.ctx, self.ring, da.ptr, db.ptr, 1, dc.ptr) });
        dc.download(&self.dev)
    }
    /// |message| mod 2^n as n bits, LSB = coefficient 0, packed LSB-first (ring_lpr.rs:424-427, 318-329)
    fn mu_bits(&self, message: impl Into<Z>) -> Vec<u8> {
        let message: Z = message.into().abs();
        let mu = message.modulo(Z::from(2).pow(&self.nz).unwrap());
        let bits = mu.to_bits();
        let mut out = vec![0u8; ((self.n + 7) / 8).max(4) as usize];
        for (i, b) in bits.iter().enumerate() {
            if *b {
                out[i / 8] |= 1 << (i % 8);
            }
        }
        out
    }
}
impl Drop for RingLPRGpu {
    fn drop(&mut self) {
        unsafe { qf_ring_destroy(self.ring) }
    }
}
impl PKEncryptionScheme for RingLPRGpu {
    type PublicKey = (PolynomialRingZq, PolynomialRingZq);
    type SecretKey = PolynomialRingZq;
    type Cipher = (PolynomialRingZq, PolynomialRingZq);

    fn gen(&self) -> (Self::PublicKey, Self::SecretKey) {
        // ring_lpr.rs:370-395: a uniform, s, e <- chi, b = a*s + e
        let (q, n, alpha) = (&self.modulus, &self.nz, &self.alpha);
        let a = PolynomialRingZq::sample_uniform(q);
        let s = PolynomialRingZq::sample_discrete_gauss(q, n, 0, alpha * &q.get_q()).unwrap();
        let e = PolynomialRingZq::sample_discrete_gauss(q, n, 0, alpha * &q.get_q()).unwrap();
        let b = u32_to_poly(&self.mul(&a, &s), q) + e; // :391
        ((a, b), s)
    }

    fn enc(&self, pk: &Self::PublicKey, message: impl Into<Z>) -> Self::Cipher {
        let (q, n, alpha) = (&self.modulus, &self.nz, &self.alpha);
        let mu = self.mu_bits(message);
        let r = PolynomialRingZq::sample_discrete_gauss(q, n, 0, alpha * &q.get_q()).unwrap(); // :430-436
        let e1 = PolynomialRingZq::sample_discrete_gauss(q, n, 0, alpha * &q.get_q()).unwrap(); // :438-444
        let e2 = PolynomialRingZq::sample_discrete_gauss(q, n, 0, alpha * &q.get_q()).unwrap(); // :446-452
        let key = self.rkey(&[&pk.0, &pk.1]);
        let d = |p: &PolynomialRingZq| self.dev.upload(&poly_to_u32(p, self.n));
        let (dr, de1, de2, dmu) = (d(&r), d(&e1), d(&e2), self.dev.upload(&mu));
        let (du, dv) = (DevBuf::<u32>::new(&self.dev, self.n as usize), DevBuf::<u32>::new(&self.dev, self.n as usize));
        // u = a*r + e1 ; v = b*r + e2 + mu*floor(q/2): one fused launch (ring_lpr.rs:455-457)
        self.dev.check(unsafe { qf_ringlpr_enc_dev(self.dev.ctx, self.ring, key, dr.ptr, de1.ptr, de2.ptr, dmu.ptr, 1, du.ptr, dv.ptr) });
        let out = (u32_to_poly(&du.download(&self.dev), q), u32_to_poly(&dv.download(&self.dev), q));
        unsafe { qf_rkey_destroy(key) };
        out
    }

    fn dec(&self, sk: &Self::SecretKey, cipher: &Self::Cipher) -> Z {
        // ring_lpr.rs:489-508: v - s*u, per-coefficient rule, Z::from_bits -- one fused launch
        let key = self.rkey(&[sk]);
        let (du, dv) = (self.dev.upload(&poly_to_u32(&cipher.0, self.n)), self.dev.upload(&poly_to_u32(&cipher.1, self.n)));
        let dmu = DevBuf::<u8>::new(&self.dev, ((self.n + 7) / 8).max(4) as usize);
        self.dev.check(unsafe { qf_ringlpr_dec_dev(self.dev.ctx, self.ring, key, du.ptr, dv.ptr, 1, dmu.ptr) });
        let bytes = dmu.download(&self.dev);
        unsafe { qf_rkey_destroy(key) };
        let bits: Vec<bool> = (0..self.n as usize).map(|i| (bytes[i / 8] >> (i % 8)) & 1 == 1).collect();
        let _ = self.q;
        Z::from_bits(&bits)
    }
}

// ------------------------------------------------------------------------------------------------
// SISHash (construction/hash/sis.rs): the reference's key field is private, so the GPU type owns its key
// ------------------------------------------------------------------------------------------------

pub struct SISHashGpu {
    key: MatZq,
    dev: Rc<Device>,
    keys: KeyCache,
}
impl SISHashGpu {
    /// sis.rs:79-93
    pub fn gen(dev: Rc<Device>, n: impl Into<Z>, m: impl Into<Z>, q: impl Into<Modulus>) -> Self {
        let (n, m, q): (Z, Z, Modulus) = (n.into(), m.into(), q.into());
        assert!(n >= Z::ONE, "n must be chosen bigger than 0.");
        Self { key: MatZq::sample_uniform(&n, &m, &q), dev, keys: KeyCache::new() }
    }
    pub fn from_key(dev: Rc<Device>, key: MatZq) -> Self {
        Self { key, dev, keys: KeyCache::new() }
    }
    /// sis.rs:161-167: `&self.key * value`
    pub fn hash(&self, value: &MatZq) -> MatZq {
        if !value.is_column_vector() {
            panic!("The hashed value has to be a column vector!");
        }
        assert!(value.get_mod() == self.key.get_mod(), "mismatching modulus");
        assert!(value.get_num_rows() == self.key.get_num_columns(), "mismatching dimensions");
        let k = self.keys.get(&self.dev, &self.key, false);
        let (v, ..) = matzq_to_u32(value);
        let mut d = vec![0u32; self.key.get_num_rows() as usize];
        self.dev.check(unsafe { qf_matvec_modq_host(self.dev.ctx, k, v.as_ptr(), d.as_mut_ptr()) });
        u32_to_matzq(&d, self.key.get_num_rows(), 1, &self.key.get_mod())
    }
    /// B bit-packed inputs in one launch (the SIS short domain): kernel (a)
    pub fn hash_batch_bits(&self, x_bits_dev: *const c_void, stride: usize, b: u64, out_dev: *mut u32) {
        let k = self.keys.get(&self.dev, &self.key, false);
        self.dev.check(unsafe { qf_matmul_modq_dev(self.dev.ctx, k, x_bits_dev, QF_X_BITS, stride, 0, b, ptr::null(), ptr::null(), out_dev) });
    }
}

// ------------------------------------------------------------------------------------------------
// GPV-ring FDH (construction/signature/fdh.rs + fdh/gpv_ring.rs): vfy on the device, gen / sign upstream
// ------------------------------------------------------------------------------------------------

type FdhRing = FDH<MatPolynomialRingZq, (MatPolyOverZ, MatPolyOverZ), MatPolyOverZ, MatPolynomialRingZq, PSFGPVRing, HashMatPolynomialRingZq>;

pub struct FdhGpvRingGpu {
    pub inner: FdhRing,
    dev: Rc<Device>,
    ring: *mut qf_ring,
    n: i64,
    q: u32,
}
impl FdhGpvRingGpu {
    /// FDH::init_gpv_ring(n, modulus, s)  (fdh/gpv_ring.rs:65-87)
    pub fn init_gpv_ring(dev: Rc<Device>, n: impl Into<Z>, modulus: impl Into<Modulus>, s: impl Into<Q>) -> Self {
        let inner = FDH::init_gpv_ring(n, modulus, s);
        let n = i64::try_from(&inner.psf.gp.n).unwrap();
        let q = modulus_u32(&inner.psf.gp.q);
        let mut ring = ptr::null_mut();
        dev.check(unsafe { qf_ring_create(dev.ctx, n as u32, q, &mut ring) });
        Self { inner, dev, ring, n, q }
    }
    /// floor(s^2 * n * (k+2)): the integer form of check_domain's bound (psf/gpv_ring.rs:266-275)
    fn norm_bound(&self, mbar: i64) -> u64 {
        let bound: Q = self.inner.psf.s.pow(2).unwrap() * Z::from(self.n * mbar);
        u64::try_from(&bound.floor()).unwrap_or(u64::MAX)
    }
}
impl Drop for FdhGpvRingGpu {
    fn drop(&mut self) {
        unsafe { qf_ring_destroy(self.ring) }
    }
}
impl SignatureScheme for FdhGpvRingGpu {
    type SecretKey = (MatPolyOverZ, MatPolyOverZ);
    type PublicKey = MatPolynomialRingZq;
    type Signature = MatPolyOverZ;

    fn gen(&mut self) -> (Self::PublicKey, Self::SecretKey) {
        self.inner.gen() // trapdoor generation: upstream (one-off, rational arithmetic)
    }
    fn sign(&mut self, m: String, sk: &Self::SecretKey, pk: &Self::PublicKey) -> Self::Signature {
        self.inner.sign(m, sk, pk) // SampleD / samp_p: upstream
    }
    /// fdh.rs:115-123: check_domain, hash(m), f_a(pk, sigma) == u -- norm, products and comparison in ONE launch
    fn vfy(&self, m: String, sigma: &Self::Signature, pk: &Self::PublicKey) -> bool {
        let mbar = i64::try_from(&self.inner.psf.gp.k).unwrap() + 2;
        if !sigma.is_column_vector() || sigma.get_num_rows() != mbar {
            return false; // the shape half of check_domain (psf/gpv_ring.rs:271-272)
        }
        // pk: 1 x mbar polynomials of R_q
        let a = pk.get_mat();
        let mut a_words = Vec::with_capacity((mbar * self.n) as usize);
        for j in 0..mbar {
            let p: PolyOverZ = a.get_entry(0, j).unwrap();
            for i in 0..self.n {
                let c: Z = p.get_coeff(i).unwrap();
                a_words.push(u32::try_from(&c.modulo(self.q)).unwrap());
            }
        }
        let mut rkey = ptr::null_mut();
        self.dev.check(unsafe { qf_rkey_upload_host(self.dev.ctx, self.ring, a_words.as_ptr(), mbar as u32, &mut rkey) });
        // sigma: mbar x n signed coefficients; anything beyond int32 is outside the domain at these parameters
        let mut sig = Vec::with_capacity((mbar * self.n) as usize);
        for j in 0..mbar {
            let p: PolyOverZ = sigma.get_entry(j, 0).unwrap();
            for i in 0..self.n {
                let c: Z = p.get_coeff(i).unwrap();
                match i32::try_from(&c) {
                    Ok(v) => sig.push(v),
                    Err(_) => {
                        unsafe { qf_rkey_destroy(rkey) };
                        return false;
                    }
                }
            }
        }
        // u = hash(m): HashMatPolynomialRingZq{1,1} (sha256.rs:249-256) -- n digests on the device
        let bytes = m.as_bytes();
        let offsets = [0u64, bytes.len() as u64];
        let (dmsg, doff) = (self.dev.upload(if bytes.is_empty() { &[0u8][..] } else { bytes }), self.dev.upload(&offsets));
        let du = DevBuf::<u32>::new(&self.dev, self.n as usize);
        self.dev.check(unsafe { qf_hash_to_mat_zq_sha256_dev(self.dev.ctx, dmsg.ptr, doff.ptr, 1, self.n as u32, 1, self.q, du.ptr) });
        let dsig = self.dev.upload(&sig);
        let dok = DevBuf::<u8>::new(&self.dev, 1);
        self.dev.check(unsafe {
            qf_fdh_ring_verify_dev(self.dev.ctx, self.ring, rkey, dsig.ptr, du.ptr, self.norm_bound(mbar), 1, ptr::null_mut(), dok.ptr)
        });
        let ok = dok.download(&self.dev)[0] == 1;
        unsafe { qf_rkey_destroy(rkey) };
        ok
    }
}

// the PSF half on its own, for callers of psf.f_a (primitive/psf/gpv_ring.rs:235-239)
impl FdhGpvRingGpu {
    pub fn f_a(&self, a: &MatPolynomialRingZq, sigma: &MatPolyOverZ) -> MatPolynomialRingZq {
        assert!(self.inner.psf.check_domain(sigma));
        let sigma_q = MatPolynomialRingZq::from((sigma, &a.get_mod())); // reduced into [0,q) first (:237)
        let mbar = a.get_num_columns();
        let mut acc = vec![0u64; self.n as usize];
        let (mat_a, mat_s) = (a.get_mat(), sigma_q.get_mat());
        for j in 0..mbar {
            let pa: PolyOverZ = mat_a.get_entry(0, j).unwrap();
            let ps: PolyOverZ = mat_s.get_entry(j, 0).unwrap();
            let words = |p: &PolyOverZ| -> Vec<u32> { (0..self.n).map(|i| { let c: Z = p.get_coeff(i).unwrap(); u32::try_from(&c.modulo(self.q)).unwrap() }).collect() };
            let (da, ds) = (self.dev.upload(&words(&pa)), self.dev.upload(&words(&ps)));
            let dc = DevBuf::<u32>::new(&self.dev, self.n as usize);
            self.dev.check(unsafe { qf_polymul_dev(self.dev.ctx, self.ring, da.ptr, ds.ptr, 1, dc.ptr) });
            for (t, c) in acc.iter_mut().zip(dc.download(&self.dev)) {
                *t = (*t + c as u64) % self.q as u64;
            }
        }
        let mut p = PolyOverZ::default();
        for (i, c) in acc.iter().enumerate() {
            p.set_coeff(i as i64, Z::from(*c)).unwrap();
        }
        let mut out = MatPolyOverZ::new(1, 1);
        out.set_entry(0, 0, p).unwrap();
        MatPolynomialRingZq::from((&out, &a.get_mod()))
    }
}

#[allow(dead_code)]
fn _type_anchors(_: Zq) {} // Zq is what dec's distance rule is stated in upstream (regev.rs:447); the device returns the bit
