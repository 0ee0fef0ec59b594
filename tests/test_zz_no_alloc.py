"""The no-allocation entry point qf_matvec_modq_static_dev (include/qfgpu.h) against the existing
qf_matvec_modq_dev and the CPU oracle, through the C ABI: two calls in a row, the second with a
smaller key, then the first key again (the shared scratch must come back clean each time).
Named test_zz_* so that pytest collects it last."""
import numpy as np
import pytest

from oracle import coracle as co
from crypto_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 0x5146414C4C


def _both(ctx, key, x_host):
    x = ctx.to_dev_u32(x_host)
    y_ref = ctx.empty_u32(key.rows)
    y_new = ctx.empty_u32(key.rows)
    y_new.fill_(-1)
    assert ctx.lib.qf_matvec_modq_dev(ctx.h, key.h, x.data_ptr(), y_ref.data_ptr()) == 0
    assert ctx.lib.qf_matvec_modq_static_dev(ctx.h, key.h, x.data_ptr(), y_new.data_ptr()) == 0
    ctx.sync()
    return ctx.to_host_u32(y_ref), ctx.to_host_u32(y_new)


def test_static_matvec_matches_existing_entry_point_and_oracle(ctx):
    shapes = [(512, 11776, 8380417),      # BASELINE config 3 key: 16 row blocks x 92 k-chunks
              (257, 6168, 16777213),      # config 2 key, smaller: 9 row blocks, a padded last block
              (51, 867, 124991),          # config 1
              (512, 11776, 8380417)]      # the large one again
    keys = {}
    for rows, cols, q in shapes:
        if (rows, cols, q) not in keys:
            A = co.sample_uniform(SEED, 1, rows, rows * cols, q).reshape(rows, cols)
            keys[(rows, cols, q)] = (A, ctx.key_upload(A, q))
    for i, (rows, cols, q) in enumerate(shapes):
        A, key = keys[(rows, cols, q)]
        x = co.sample_uniform(SEED, 6, 1000 * i, cols, q)
        y_ref, y_new = _both(ctx, key, x)
        want = co.matvec(A, x, q)
        assert np.array_equal(y_ref, want), (i, "existing entry point vs oracle")
        assert np.array_equal(y_new, want), (i, "static entry point vs oracle")
        assert np.array_equal(y_new, y_ref)


def test_static_matvec_rejects_oversize_and_foreign_arguments(ctx):
    q = 257
    A = co.sample_uniform(SEED, 1, 0, 32800 * 4, q).reshape(32800, 4)      # 32800 rows > 32768
    key = ctx.key_upload(A, q)
    x = ctx.to_dev_u32(np.ones(4, np.uint32))
    y = ctx.empty_u32(32800)
    assert ctx.lib.qf_matvec_modq_static_dev(ctx.h, key.h, x.data_ptr(), y.data_ptr()) == _lib.QF_ERR_NOTSUP
    assert b"32768" in ctx.lib.qf_last_error(ctx.h)
    assert ctx.lib.qf_matvec_modq_dev(ctx.h, key.h, x.data_ptr(), y.data_ptr()) == 0   # the allocating one still serves it
    ctx.sync()
    assert np.array_equal(ctx.to_host_u32(y), co.matvec(A, np.ones(4, np.uint32), q))
    assert ctx.lib.qf_matvec_modq_static_dev(ctx.h, key.h, None, y.data_ptr()) == _lib.QF_ERR_ARG


def test_static_matvec_allocates_nothing(ctx):
    """The property the entry point exists for: no hipMalloc / hipFree on the call path.  The device's free
    memory (hipMemGetInfo, which sees every allocation of the process, the library's included) is the same
    before and after a series of static calls on keys of different sizes -- while the allocating entry point,
    called first on a fresh context, does move it (its cross-workgroup sums are allocated on first use)."""
    import torch
    from crypto_amd.device import Context
    c2 = Context(0)                                             # fresh context: nothing pooled yet
    q = 8380417
    A = co.sample_uniform(SEED, 1, 7, 512 * 11776, q).reshape(512, 11776)
    big, small = c2.key_upload(A, q), c2.key_upload(A[:100, :900].copy(), q)
    xb, xs = c2.to_dev_u32(co.sample_uniform(SEED, 6, 0, 11776, q)), c2.to_dev_u32(co.sample_uniform(SEED, 6, 0, 900, q))
    yb, ys = c2.empty_u32(512), c2.empty_u32(100)
    c2.sync()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(3):
        assert c2.lib.qf_matvec_modq_static_dev(c2.h, big.h, xb.data_ptr(), yb.data_ptr()) == 0
        assert c2.lib.qf_matvec_modq_static_dev(c2.h, small.h, xs.data_ptr(), ys.data_ptr()) == 0
    c2.sync()
    assert torch.cuda.mem_get_info(0)[0] == free0
    assert np.array_equal(c2.to_host_u32(yb), co.matvec(A, c2.to_host_u32(xb), q))
    assert np.array_equal(c2.to_host_u32(ys), co.matvec(A[:100, :900], c2.to_host_u32(xs), q))


def test_two_devices_replicate_and_reject(ctx):
    """Needs two GPUs (skipped on the one-GPU box): qf_key_replicate's peer copy onto a second device gives a key
    that computes the same product there, and every entry point refuses a key that lives on another device."""
    import ctypes as C
    import torch
    from crypto_amd.device import Context, Key
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    c1 = Context(1)
    q = 16777213
    A = co.sample_uniform(SEED, 1, 0, 257 * 700, q).reshape(257, 700)
    k0 = ctx.key_upload(A, q)
    ctxs = (C.c_void_p * 1)(c1.h)
    out = (C.c_void_p * 1)()
    assert ctx.lib.qf_key_replicate(ctxs, 1, k0.h, out) == 0
    k1 = Key(c1, C.c_void_p(out[0]))
    x = co.sample_uniform(SEED, 6, 0, 700, q)
    assert np.array_equal(c1.matvec(k1, x), co.matvec(A, x, q))
    xd, yd = c1.to_dev_u32(x), c1.empty_u32(257)
    for fn in (c1.lib.qf_matvec_modq_dev, c1.lib.qf_matvec_modq_static_dev):
        assert fn(c1.h, k0.h, xd.data_ptr(), yd.data_ptr()) == _lib.QF_ERR_ARG      # k0 lives on device 0
    torch.cuda.set_device(0)
