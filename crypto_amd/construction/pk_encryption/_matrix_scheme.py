"""Shared plumbing of the three matrix schemes (Regev, DualRegev, LPR): device key caching,
single-message products on kernel (c), batched products on kernel (a), batched decryption."""
from __future__ import annotations

import numpy as np

from ...device import X_BITS, X_U32, bit_stride, pack_bits
from ...math import MatZ, MatZq
from ...runtime import Randomness, default_context


class MatrixSchemeMixin:
    def _init_runtime(self, ctx=None, seed=None, noise_model=None):
        self._ctx = ctx
        self.rng = Randomness(seed, ctx)   # seed None -> a fresh os.urandom seed per object
        set_noise_model(self, noise_model)

    @property
    def ctx(self):
        return self._ctx or default_context()

    def _dev_key(self, mat: MatZq, transpose: bool = False):
        """Upload (once) and cache the digit planes of a public matrix."""
        tag = (id(self.ctx), transpose)
        key = mat._dev.get(tag)
        if key is None or key.h is None:
            key = self.ctx.key_upload(mat.entries, mat.q, transpose=transpose)
            mat._dev[tag] = key
        return key

    def _lwe_keypair(self, n: int, m: int, secret_from_chi: bool):
        """pk = [A ; s^t A + e^t], sk = s (regev.rs:350-374 with s uniform, lpr.rs:338-370 with s <- chi).
        Everything is drawn and multiplied on the device, straight into one image [pk | s]; the host
        gets one copy of it, and pk's digit planes are built from the device image (so the first enc
        under this key uploads nothing)."""
        q, ctx, rng, W = self.q, self.ctx, self.rng, self.noise_width
        img = ctx.empty_u32((n + 1) * m + n)
        pk_dev, s_dev = img[:(n + 1) * m].view(n + 1, m), img[(n + 1) * m:]
        ctx.sample_uniform(rng.seed, 1, rng._take(1, n * m), n * m, q, out=pk_dev[:n])
        if secret_from_chi:
            s_dev.copy_(rng.noise_dev(2, n, q, W))
        else:
            ctx.sample_uniform(rng.seed, 2, rng._take(2, n), n, q, out=s_dev)
        e_dev = rng.noise_dev(4, m, q, W)
        ctx.vecmat_u32(pk_dev[:n], q, s_dev, addend=e_dev, out=pk_dev[n])
        host = ctx.to_host_u32(img)
        pk = MatZq.from_canonical(host[:(n + 1) * m].reshape(n + 1, m), q)
        pk._dev[(id(ctx), False)] = ctx.key_upload(pk_dev, q)
        return pk, MatZq.from_canonical(host[(n + 1) * m:].reshape(n, 1), q)

    def _check_pk(self, pk: MatZq, rows: int, cols: int):
        # the reference panics on mismatching dimensions / moduli inside qfall-math
        if not isinstance(pk, MatZq):
            raise TypeError("public key must be a MatZq")
        if pk.q != self.q:
            raise ValueError("mismatching modulus between scheme and key")
        if pk.entries.shape != (rows, cols):
            raise ValueError(f"mismatching matrix dimensions: expected {rows}x{cols}, got {pk.entries.shape}")

    def _decode_batch(self, w: np.ndarray, ciphers) -> np.ndarray:
        """bits = decision rule applied to <w, c[:-1]> + c[-1] for each row of `ciphers`."""
        if isinstance(ciphers, np.ndarray):
            # weights and ciphertexts cross PCIe as ONE buffer (a single-ciphertext call is all latency)
            c2 = np.ascontiguousarray(ciphers, dtype=np.uint32)
            both = self.ctx.to_dev_u32(np.concatenate([np.asarray(w, np.uint32).reshape(-1), c2.reshape(-1)]))
            wd, c = both[:w.size], both[w.size:].view(c2.shape)
        else:
            wd, c = self.ctx.to_dev_u32(w), ciphers
        bits = self.ctx.dot_decode(wd, self.q, c)
        return self.ctx.to_host_u8(bits)


def set_noise_model(scheme, noise_model):
    """noise_model: None / "discrete_gauss" (default: D_{Z, alpha*q} with security parameter n, as the
    reference samples chi -- regev.rs:356-363, lpr.rs:342-360, ring_lpr.rs:374-388) or "irwin_hall"
    (the bounded stand-in of the benchmark generator, SURVEY 8d: explicit opt-in)."""
    if noise_model == "irwin_hall":
        scheme.rng.gauss = None
    elif noise_model in (None, "discrete_gauss"):
        scheme.rng.gauss = (int(scheme.n), float(scheme.alpha * getattr(scheme, "q_int", scheme.q)))
    else:
        raise ValueError("noise_model must be 'irwin_hall' or 'discrete_gauss'")


def message_bit(message) -> int:
    """message.modulo(2): Euclidean remainder, also for negative values (regev.rs:398,598-612)."""
    return int(message) % 2
