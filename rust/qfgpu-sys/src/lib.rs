//! Raw `extern "C"` declarations of `include/qfgpu.h` -- one per symbol, same order, same names.
//! UNCOMPILED in this repository (no Rust toolchain in the build image); kept in lock-step with the
//! header by `tests/test_host_logic.py::test_rust_sys_crate_matches_the_header`.
//! Conventions, limits and the reference call site each entry point replaces: see the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct qf_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qf_key {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qf_ring {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qf_rkey {
    _private: [u8; 0],
}

pub const QF_OK: c_int = 0;
pub const QF_ERR_ARG: c_int = 1;
pub const QF_ERR_DIM: c_int = 2;
pub const QF_ERR_MODULUS: c_int = 3;
pub const QF_ERR_DEVICE: c_int = 4;
pub const QF_ERR_RANGE: c_int = 5;
pub const QF_ERR_NOTSUP: c_int = 6;

pub const QF_X_BITS: c_int = 0;
pub const QF_X_U32: c_int = 1;
pub const QF_X_I32: c_int = 2;
pub const QF_X_BITS_KT: c_int = 3;
pub const QF_X_PLANES: c_int = 4;

extern "C" {
    // ---- context
    pub fn qf_version() -> *const c_char;
    pub fn qf_device_count(count: *mut c_int) -> c_int;
    pub fn qf_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut qf_ctx) -> c_int;
    pub fn qf_ctx_set_stream(ctx: *mut qf_ctx, stream: *mut c_void) -> c_int;
    pub fn qf_ctx_destroy(ctx: *mut qf_ctx);
    pub fn qf_sync(ctx: *mut qf_ctx) -> c_int;
    pub fn qf_last_error(ctx: *const qf_ctx) -> *const c_char;
    pub fn qf_malloc(ctx: *mut qf_ctx, bytes: usize, dev_out: *mut *mut c_void) -> c_int;
    pub fn qf_free(ctx: *mut qf_ctx, dev: *mut c_void) -> c_int;
    pub fn qf_memcpy_h2d(ctx: *mut qf_ctx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> c_int;
    pub fn qf_memcpy_d2h(ctx: *mut qf_ctx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> c_int;

    // ---- Z_q matrix keys
    pub fn qf_key_upload_host(ctx: *mut qf_ctx, a_host: *const u32, rows: u32, cols: u32, q: u32, transpose: c_int, out: *mut *mut qf_key) -> c_int;
    pub fn qf_key_upload_dev(ctx: *mut qf_ctx, a_dev: *const u32, rows: u32, cols: u32, q: u32, transpose: c_int, out: *mut *mut qf_key) -> c_int;
    pub fn qf_key_destroy(key: *mut qf_key);
    pub fn qf_key_dims(key: *const qf_key, rows: *mut u32, cols: *mut u32, q: *mut u32) -> c_int;
    pub fn qf_key_u32_dev(key: *const qf_key, dev_ptr: *mut *const u32, bytes: *mut usize) -> c_int;
    pub fn qf_key_replicate(ctxs: *const *mut qf_ctx, n: c_int, src_key: *const qf_key, out_keys: *mut *mut qf_key) -> c_int;

    // ---- kernel (a): batched product on int8 MFMA
    pub fn qf_matmul_modq_dev(ctx: *mut qf_ctx, key: *const qf_key, x_dev: *const c_void, x_kind: c_int, x_stride_bytes: usize, x_digits: c_int, b: u64, addend_dev: *const u32, msg_dev: *const u8, out_dev: *mut u32) -> c_int;
    pub fn qf_operand_planes_bytes(key: *const qf_key, b: u64, digits: c_int) -> usize;
    pub fn qf_recode_planes_dev(ctx: *mut qf_ctx, key: *const qf_key, x_dev: *const c_void, x_kind: c_int, x_stride_bytes: usize, digits: c_int, b: u64, planes_dev: *mut i8) -> c_int;
    pub fn qf_bits_to_kt_dev(ctx: *mut qf_ctx, bits_dev: *const u8, stride_bytes: usize, b: u64, cols: u32, kt_dev: *mut u8, kt_stride_bytes: usize) -> c_int;

    // ---- kernel (c): single-message products
    pub fn qf_matvec_modq_host(ctx: *mut qf_ctx, key: *const qf_key, x_host: *const u32, y_host: *mut u32) -> c_int;
    pub fn qf_vecmat_modq_host(ctx: *mut qf_ctx, key: *const qf_key, x_host: *const u32, y_host: *mut u32) -> c_int;
    pub fn qf_matvec_modq_dev(ctx: *mut qf_ctx, key: *const qf_key, x_dev: *const u32, y_dev: *mut u32) -> c_int;
    pub fn qf_vecmat_modq_dev(ctx: *mut qf_ctx, key: *const qf_key, x_dev: *const u32, y_dev: *mut u32) -> c_int;
    pub fn qf_vecmat_u32_dev(ctx: *mut qf_ctx, a_dev: *const u32, rows: u32, cols: u32, q: u32, x_dev: *const u32, addend_dev: *const u32, y_dev: *mut u32) -> c_int;
    pub fn qf_matvec_modq_static_dev(ctx: *mut qf_ctx, key: *const qf_key, x_dev: *const u32, y_dev: *mut u32) -> c_int;

    // ---- batched decryption
    pub fn qf_dot_decode_dev(ctx: *mut qf_ctx, w_dev: *const u32, n: u32, q: u32, c_dev: *const u32, b: u64, bits_dev: *mut u8) -> c_int;

    // ---- R_q
    pub fn qf_ring_create(ctx: *mut qf_ctx, n: u32, q: u32, out: *mut *mut qf_ring) -> c_int;
    pub fn qf_ring_destroy(ring: *mut qf_ring);
    pub fn qf_ring_info(ring: *const qf_ring, n: *mut u32, q: *mut u32, uses_ntt: *mut c_int) -> c_int;
    pub fn qf_rkey_upload_host(ctx: *mut qf_ctx, ring: *const qf_ring, polys_host: *const u32, npoly: u32, out: *mut *mut qf_rkey) -> c_int;
    pub fn qf_rkey_upload_dev(ctx: *mut qf_ctx, ring: *const qf_ring, polys_dev: *const u32, npoly: u32, out: *mut *mut qf_rkey) -> c_int;
    pub fn qf_rkey_destroy(rkey: *mut qf_rkey);
    pub fn qf_polymul_dev(ctx: *mut qf_ctx, ring: *const qf_ring, a_dev: *const u32, b_dev: *const u32, b: u64, c_dev: *mut u32) -> c_int;
    pub fn qf_ringlpr_enc_dev(ctx: *mut qf_ctx, ring: *const qf_ring, pk: *const qf_rkey, r_dev: *const u32, e1_dev: *const u32, e2_dev: *const u32, mu_bits_dev: *const u8, b: u64, u_dev: *mut u32, v_dev: *mut u32) -> c_int;
    pub fn qf_ringlpr_dec_dev(ctx: *mut qf_ctx, ring: *const qf_ring, sk: *const qf_rkey, u_dev: *const u32, v_dev: *const u32, b: u64, mu_bits_dev: *mut u8) -> c_int;
    pub fn qf_fdh_ring_verify_dev(ctx: *mut qf_ctx, ring: *const qf_ring, a: *const qf_rkey, sigma_dev: *const i32, u_dev: *const u32, norm_bound: u64, b: u64, t_dev: *mut u32, ok_dev: *mut u8) -> c_int;
    pub fn qf_hash_to_mat_zq_sha256_dev(ctx: *mut qf_ctx, msgs_dev: *const u8, offsets_dev: *const u64, b: u64, rows: u32, cols: u32, q: u32, out_dev: *mut u32) -> c_int;

    // ---- counter-mode generator: seeded (SplitMix64) and keyed (ChaCha20)
    pub fn qf_sample_uniform_dev(ctx: *mut qf_ctx, seed: u64, stream: u32, first: u64, count: u64, q: u32, out_dev: *mut u32) -> c_int;
    pub fn qf_sample_bits_dev(ctx: *mut qf_ctx, seed: u64, stream: u32, first_row: u64, rows: u64, cols: u32, stride_bytes: u32, out_dev: *mut u8) -> c_int;
    pub fn qf_sample_bits_kt_dev(ctx: *mut qf_ctx, seed: u64, stream: u32, first_row: u64, rows: u64, cols: u32, kt_stride_bytes: usize, out_dev: *mut u8) -> c_int;
    pub fn qf_sample_noise_dev(ctx: *mut qf_ctx, seed: u64, stream: u32, first: u64, count: u64, q: u32, w: u32, out_dev: *mut u32) -> c_int;
    pub fn qf_sample_discrete_gauss_dev(ctx: *mut qf_ctx, seed: u64, stream: u32, first: u64, count: u64, q: u32, n: u32, s: f64, out_dev: *mut u32) -> c_int;
    pub fn qf_sample_uniform_keyed_dev(ctx: *mut qf_ctx, key32: *const u8, nonce: u64, stream: u32, first: u64, count: u64, q: u32, out_dev: *mut u32) -> c_int;
    pub fn qf_sample_bits_keyed_dev(ctx: *mut qf_ctx, key32: *const u8, nonce: u64, stream: u32, first_row: u64, rows: u64, cols: u32, stride_bytes: u32, out_dev: *mut u8) -> c_int;
    pub fn qf_sample_bits_kt_keyed_dev(ctx: *mut qf_ctx, key32: *const u8, nonce: u64, stream: u32, first_row: u64, rows: u64, cols: u32, kt_stride_bytes: usize, out_dev: *mut u8) -> c_int;
    pub fn qf_sample_noise_keyed_dev(ctx: *mut qf_ctx, key32: *const u8, nonce: u64, stream: u32, first: u64, count: u64, q: u32, w: u32, out_dev: *mut u32) -> c_int;
    pub fn qf_sample_discrete_gauss_keyed_dev(ctx: *mut qf_ctx, key32: *const u8, nonce: u64, stream: u32, first: u64, count: u64, q: u32, n: u32, s: f64, out_dev: *mut u32) -> c_int;

    // ---- introspection for benches
    pub fn qf_set_timing(ctx: *mut qf_ctx, enabled: c_int) -> c_int;
    pub fn qf_last_kernel_ms(ctx: *mut qf_ctx, which: *const c_char, ms: *mut f32) -> c_int;
}
