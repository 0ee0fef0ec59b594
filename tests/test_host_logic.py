"""CPU: host-side logic that needs no GPU -- the C ABI library loads and exports every symbol
include/qfgpu.h declares, parameter generation follows the reference's shape rules, value types
parse/print the qfall text formats."""
import os
import re

import numpy as np
import pytest

from crypto_amd import _lib
from crypto_amd.math import MatPolyOverZ, MatZ, MatZq, MathError, format_poly, gaussian_width, is_prime, parse_poly
from crypto_amd.construction.pk_encryption import LPR, DualRegev, Regev, RingLPR
from crypto_amd.construction.hash.sha256 import hash_to_mat_zq_sha256, hash_to_zq_sha256, sha256
from crypto_amd.construction.hash import SISHash
from crypto_amd.primitive.psf import GadgetParametersRing, PSFGPVRing
from crypto_amd.utils import new_anticyclic, new_cyclic, rot_minus, rot_minus_matrix
from crypto_amd.device import bit_stride, pack_bits, unpack_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # raises if the .so is missing or lacks a symbol listed in _lib.SYMBOLS
    header = open(os.path.join(ROOT, "include", "qfgpu.h")).read()
    declared = set(re.findall(r"\b(qf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.qf_version().startswith(b"qfgpu")


def _c_to_rust(ctype: str) -> str:
    """One C parameter / return type of include/qfgpu.h as the Rust FFI type that must stand for it."""
    t = re.sub(r"/\*.*?\*/", "", ctype).strip()
    base = {"int": "c_int", "void": "c_void", "char": "c_char", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int8_t": "i8", "int32_t": "i32",
            "size_t": "usize", "double": "f64", "float": "f32", "qf_ctx": "qf_ctx", "qf_key": "qf_key", "qf_ring": "qf_ring", "qf_rkey": "qf_rkey"}
    toks = re.findall(r"[A-Za-z_0-9]+|\*", t)
    # C declarators read right to left: "const T *" = pointer to const T; "T *const *" = pointer to const pointer to T
    const_base = toks[0] == "const"
    if const_base:
        toks = toks[1:]
    out = base[toks[0]]
    inner_const = const_base
    for tok in toks[1:]:
        if tok == "*":
            out = ("*const " if inner_const else "*mut ") + out
            inner_const = False
        elif tok == "const":
            inner_const = True        # qualifies the pointer just formed: the NEXT '*' points to a const pointer
        else:
            raise AssertionError(f"unexpected token {tok!r} in C type {ctype!r}")
    return out


def test_rust_sys_crate_matches_the_header():
    """rust/qfgpu-sys/src/lib.rs (uncompiled: no Rust toolchain here) declares EVERY function of include/qfgpu.h with
    the same name, the same number of parameters and, parameter by parameter, the Rust type the C type maps to; the
    constants agree; and the trait-impl crate calls nothing the -sys crate does not declare."""
    header = open(os.path.join(ROOT, "include", "qfgpu.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    cdecl = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*(qf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.M | re.S):
        params = [] if args.strip() == "void" else [a.strip() for a in args.replace("\n", " ").split(",")]
        ptypes = [re.sub(r"\s*[A-Za-z_0-9]+$", "", a) if not a.rstrip().endswith("*") else a for a in params]   # drop the parameter name
        cdecl[name] = (ret.strip(), ptypes)
    assert set(cdecl) == set(_lib.SYMBOLS), set(cdecl) ^ set(_lib.SYMBOLS)
    rs = open(os.path.join(ROOT, "rust", "qfgpu-sys", "src", "lib.rs")).read()
    rdecl = {}
    for name, args, ret in re.findall(r"pub fn (qf_[a-z0-9_]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rs):
        ptypes = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        rdecl[name] = ((ret or "").strip(), ptypes)
    assert set(rdecl) == set(cdecl), set(rdecl) ^ set(cdecl)
    for name, (cret, cparams) in cdecl.items():
        rret, rparams = rdecl[name]
        assert len(rparams) == len(cparams), (name, cparams, rparams)
        for i, (c, r) in enumerate(zip(cparams, rparams)):
            assert _c_to_rust(c) == r, (name, i, c, r)
        assert (cret == "void" and rret == "") or _c_to_rust(cret) == rret, (name, cret, rret)
    for cname, val in re.findall(r"(QF_[A-Z0-9_]+)\s*=\s*(\d+)", header):
        assert re.search(rf"pub const {cname}: c_int = {val};", rs), cname
    gpu = open(os.path.join(ROOT, "rust", "qfall-crypto-gpu", "src", "lib.rs")).read()
    called = set(re.findall(r"\b(qf_[a-z0-9_]+)\(", gpu))
    assert called <= set(cdecl), called - set(cdecl)
    for entry in ("qf_matmul_modq_dev", "qf_matvec_modq_host", "qf_vecmat_modq_host", "qf_dot_decode_dev", "qf_ringlpr_enc_dev",
                  "qf_ringlpr_dec_dev", "qf_fdh_ring_verify_dev", "qf_hash_to_mat_zq_sha256_dev", "qf_polymul_dev"):
        assert entry in called, entry
    for impl in ("impl PKEncryptionScheme for RegevGpu", "impl PKEncryptionScheme for DualRegevGpu", "impl PKEncryptionScheme for LPRGpu",
                 "impl PKEncryptionScheme for RingLPRGpu", "impl SignatureScheme for FdhGpvRingGpu", "pub fn hash(&self, value: &MatZq) -> MatZq"):
        assert impl in gpu, impl


def test_no_product_module_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "crypto_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", "") or "import oracle" not in src and "from oracle" not in src, f
                assert "from oracle" not in src and "import oracle" not in src, f


def test_regev_parameter_rules():
    """regev.rs:174-201, 229-295 and Appendix A of SURVEY.md."""
    for n in (10, 20, 50, 60):
        r = Regev.new_from_n(n)
        assert is_prime(r.q) and -(-n ** 3 // 2) <= r.q <= n ** 3
        assert r.m == (n + 1) * (r.q - 1).bit_length()
        r.check_correctness(); r.check_security()
    with pytest.raises(ValueError):
        Regev.new_from_n(9)
    d = Regev.default()
    assert (d.n, d.m, d.q, float(d.alpha)) == (13, 154, 1427, 0.01)
    d.check_correctness(); d.check_security()
    s = Regev.secure128()
    assert (s.n, s.m, s.q) == (230, 5313, 7764299)
    with pytest.raises(MathError):
        Regev(13, 50, 1427, 0.01).check_security()      # m <= (n+1) log q
    with pytest.raises(MathError):
        Regev(13, 154, 1427, 0.5).check_correctness()   # alpha too large
    with pytest.raises(ValueError):
        Regev(3, 16, 1, 2)                               # modulus <= 1
    assert Regev.gen_new_public_parameters(3)[1] <= 3 ** 5


def test_lpr_and_ring_lpr_parameter_rules():
    for n in (10, 64, 200):
        l = LPR.new_from_n(n)
        assert is_prime(l.q) and -(-n ** 3 // 2) <= l.q <= n ** 3
        l.check_correctness(); l.check_security()
    assert (LPR.default().n, LPR.default().q) == (10, 983)
    assert (LPR.secure128().n, LPR.secure128().q) == (500, 76859609)
    r = RingLPR.new_from_n(64)
    r.check_correctness(); r.check_security()
    assert r.q.get_degree() == 64 and r.q.is_anticyclic()
    with pytest.raises(MathError):
        RingLPR(12, 2399, 0.0039).check_correctness()   # n not a power of two
    assert (RingLPR.default().n, RingLPR.default().q_int) == (16, 2399)
    assert (RingLPR.secure128().n, RingLPR.secure128().q_int) == (512, 92897729)
    assert DualRegev.default().m == 154


def test_ring_message_encoding():
    r = RingLPR.default()
    assert r.mu_bits(65536).tolist() == [0, 0]                       # ring_lpr.rs:655-667
    assert r.mu_bits(-5).tolist() == [5, 0]                          # |message|
    assert unpack_bits(r.mu_bits(212)[None], 16)[0].tolist() == [(212 >> i) & 1 for i in range(16)]


def test_sis_checks_without_gpu():
    key = MatZq(np.zeros((5, 18), np.uint32), 11)
    SISHash(key).check_security()                                    # sis.rs doc example (5,18,11)
    with pytest.raises(MathError):
        SISHash(MatZq(np.zeros((5, 10), np.uint32), 11)).check_security()
    with pytest.raises(MathError):
        SISHash.gen(0, 3, 7)
    with pytest.raises(ValueError):
        SISHash(key).hash(MatZq(np.zeros((1, 18), np.uint32), 11))  # not a column vector (sis.rs:237-265)


def test_text_formats():
    m = MatZq.from_str("[[1, 2],[3, 12]] mod 7")
    assert m.entries.tolist() == [[1, 2], [3, 5]] and repr(m) == "[[1, 2],[3, 5]] mod 7"
    assert MatZq.from_str("[[1],[2],[3]] mod 7").is_column_vector()
    assert MatZ.from_str("[[1],[5],[-1],[9]]").entries[:, 0].tolist() == [1, 5, -1, 9]
    assert parse_poly("4  2 8 8 12") == [2, 8, 8, 12] and parse_poly("0") == []
    assert format_poly([2, 8, 8, 12, 0]) == "4  2 8 8 12" and format_poly([0, 0]) == "0"
    s = MatPolyOverZ.from_str("[[4  -1 7 6 -8, 3  0 -2 4, 4  0 3 -4 1, 4  6 4 -1 3]]", 4)
    assert s.entries.tolist() == [[-1, 7, 6, -8], [0, -2, 4, 0], [0, 3, -4, 1], [6, 4, -1, 3]]


def test_utils_match_reference_kats():
    assert rot_minus([[1], [5], [-1], [9]]).tolist() == [[1, -9, 1, -5], [5, 1, -9, 1], [-1, 5, 1, -9], [9, -1, 5, 1]]
    assert rot_minus([[1, 5, -1, 9]]).tolist() == rot_minus([1, 5, -1, 9]).tolist()
    with pytest.raises(ValueError):
        rot_minus([[1, 5, -1, 9], [1, 2, 3, 4]])
    u = 2 ** 64 - 1
    assert rot_minus_matrix([[1, 5, -1, 9], [u, 1, 2, 3]]).tolist() == [[1, -u, 5, -1, -1, -2, 9, -3], [u, 1, 1, 5, 2, -1, 3, 9]]
    assert new_anticyclic(4, 17).coeffs.tolist() == [1, 0, 0, 0, 1]     # common_moduli.rs:99-111
    assert new_cyclic(3, 7).coeffs.tolist() == [6, 0, 0, 1]
    with pytest.raises(ValueError):
        new_anticyclic(0, 7)


def test_sha_hashes_host():
    assert sha256("Hello World!") == "7f83b1657ff1fc53b92dc18148a1d65dfc2d4b1fa3d677284addd200126d9069"
    assert hash_to_zq_sha256("Hello World!", 7) == 2 and hash_to_zq_sha256("qfall", 16) == 12
    assert hash_to_mat_zq_sha256("Hello World!", 2, 2, 256) == MatZq.from_str("[[159, 26],[249, 141]] mod 256")
    with pytest.raises(ValueError):
        hash_to_mat_zq_sha256("x", 0, 2, 7)


def test_gadget_parameters_and_norm_bound():
    gp = GadgetParametersRing.init_default(8, 512)
    assert gp.k == 9 and gp.modulus.get_degree() == 8
    psf = PSFGPVRing(gp, 100)
    assert psf.mbar == 11 and psf.norm_bound() == 100 * 100 * 8 * 11
    assert GadgetParametersRing(1024, 1073479681).k == 30


def test_bit_packing_helpers():
    rng = np.random.default_rng(0)
    b = rng.integers(0, 2, (5, 867)).astype(np.uint8)
    p = pack_bits(b)
    assert p.shape == (5, bit_stride(867)) and bit_stride(867) % 16 == 0
    assert np.array_equal(unpack_bits(p, 867), b)
    assert gaussian_width(0) == 0 and gaussian_width(6711) == 2319


def test_gaussian_regularity_new_from_n_reports_empty_ranges():
    """For n = 31..36 no prime of [n^3/2, n^3] satisfies q*alpha >= n (regev_discrete_gauss.rs:161-199, 270-297):
    the reference's new_from_n would loop forever; the mirror raises MathError.  Neighbouring n still work."""
    from crypto_amd.construction.pk_encryption import RegevWithDiscreteGaussianRegularity as R
    from crypto_amd.math import MathError
    with pytest.raises(MathError):
        R.new_from_n(33, ctx=object())
    for n in (30, 37):
        p = R.new_from_n(n, ctx=object())
        p.check_correctness(); p.check_security()


def test_asm_window_checker_flags_a_touched_destination(tmp_path):
    """crypto_amd/csrc/check_asm_window.py (build guard of qf_gemm's inline-asm operand loads): a compiler
    instruction that names a load's destination register before the matching asm wait fails the build."""
    import subprocess, sys
    tool = os.path.join(ROOT, "crypto_amd", "csrc", "check_asm_window.py")
    body = ("_Z7qf_gemmv: ; @_Z7qf_gemmv\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
            "%s\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v5, v11\n\ts_endpgm\n.Lfunc_end0:\n")
    good, bad = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(body % "\tv_add_u32_e32 v4, v2, v3\n")
    bad.write_text(body % "\tv_accvgpr_write_b32 a0, v12\n")
    assert subprocess.run([sys.executable, tool, str(good), "qf_gemm"], capture_output=True).returncode == 0
    r = subprocess.run([sys.executable, tool, str(bad), "qf_gemm"], capture_output=True, text=True)
    assert r.returncode != 0 and "v[12]" in r.stdout
    # the state follows branches, not file order: a path that jumps over the wait reaches the use with the LDS
    # read still in flight
    cfg = ("_Z7qf_gemmv: ; @_Z7qf_gemmv\n\t;;#ASMSTART\n\tds_read_b128 v[20:23], v1\n\t;;#ASMEND\n\ts_cbranch_scc1 .LBB0_2\n"
           "; %%bb.1:\n\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n%s.LBB0_2:\n\tv_mov_b32_e32 v6, v21\n\ts_endpgm\n.Lfunc_end0:\n")
    bad2 = tmp_path / "bad2.s"
    bad2.write_text(cfg % "")
    r = subprocess.run([sys.executable, tool, str(bad2), "qf_gemm"], capture_output=True, text=True)
    assert r.returncode != 0 and "v[21]" in r.stdout


def test_fp64_epilogue_arithmetic_is_exact():
    """The arithmetic of qf_epi_f64 (crypto_amd/csrc/qf_gemm.h), restated in numpy float64 and checked against
    Python integers at the bounds the launcher allows: accumulators up to 2^31 - 1 in magnitude, q up to 2^31 - 1,
    one-step recombination while 257 * 65536 * |acc| + 2^34 < 2^52, the two-step form beyond."""
    rng = np.random.default_rng(5)

    def epi(a, extra, q, two_step):
        qd, qinv = np.float64(q), np.float64(1.0) / np.float64(q)
        x = a[0].astype(np.float64) + extra
        x = a[1].astype(np.float64) * 256.0 + x              # each fma below is exact, so mul + add equals it
        hi = a[3].astype(np.float64) * 256.0 + a[2].astype(np.float64)
        if two_step:
            hi = hi - np.rint(hi * qinv) * qd
        x = hi * 65536.0 + x
        t = np.rint(x * qinv)
        r = (x - t * qd).astype(np.int64)                    # fma(-t, q, x): t*q < 2^53, the difference is exact
        return np.where(r < 0, r + q, r)

    for q in (2, 3, 257, 16777213, 1073741789, 2147483629, 2147483647):
        for amax, two_step in ((2**25, False), (2**27 - 1, False), (2**31 - 1, True), (2**20, True)):
            if not two_step:
                assert amax * 257 * 65536 + 2**34 < 2**52
            a = rng.integers(-amax, amax + 1, size=(4, 4000), dtype=np.int64)
            a[:, 0], a[:, 1] = amax, -amax                   # the corners
            extra = rng.integers(0, 2 * q, size=4000).astype(np.float64)
            got = epi(a, extra, q, two_step)
            want = [(int(a[0, i]) + 256 * int(a[1, i]) + 65536 * (int(a[2, i]) + 256 * int(a[3, i])) + int(extra[i])) % q for i in range(4000)]
            assert got.tolist() == want, (q, amax, two_step)
