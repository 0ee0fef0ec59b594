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
