"""Full-domain-hash signatures -- verify path of construction/signature/fdh.rs with the ring
instantiation of signature/fdh/gpv_ring.rs.  `gen` / `sign` need the trapdoor sampler and are
out of scope; `vfy` is the hot path."""
from __future__ import annotations

import numpy as np

from ...math import MatPolyOverZ
from ...primitive.psf.gpv import GadgetParameters, PSFGPV
from ...primitive.psf.gpv_ring import GadgetParametersRing, PSFGPVRing
from ..hash.sha256 import HashMatPolynomialRingZq, HashMatZq


class FDH:
    def __init__(self, psf: PSFGPVRing, hash_obj: HashMatPolynomialRingZq):
        self.psf, self.hash = psf, hash_obj
        self.storage = {}

    @classmethod
    def init_gpv_ring(cls, n, modulus, s, ctx=None) -> "FDH":
        """fdh/gpv_ring.rs:65-87."""
        psf = PSFGPVRing(GadgetParametersRing.init_default(n, modulus), s, ctx=ctx)
        return cls(psf, HashMatPolynomialRingZq(psf.gp.modulus, 1, 1))

    @classmethod
    def init_gpv(cls, n, modulus, s, ctx=None) -> "FDH":
        """fdh/gpv.rs:40-60: the classical instantiation, hash into Z_q^{n x 1}."""
        psf = PSFGPV(GadgetParameters.init_default(n, modulus), s, ctx=ctx)
        return cls(psf, HashMatZq(int(modulus), int(n), 1))

    def gen(self):
        return self.psf.trap_gen()

    def sign(self, m: str, sk, pk):
        raise NotImplementedError("signing needs the trapdoor preimage sampler (out of scope)")

    def vfy(self, m: str, sigma: MatPolyOverZ, pk) -> bool:
        """fdh.rs:115-123: check_domain(sigma) && f_a(pk, sigma) == hash(m)."""
        return self.vfy_batch([m], [sigma], pk)[0]

    def vfy_batch(self, messages, sigmas, pk) -> list:
        """B verifications in one launch.  pk: (k+2, n) residues."""
        psf = self.psf
        B = len(messages)
        if isinstance(psf, PSFGPV):
            # classical instantiation: domain test on the host (exact), hash and A*sigma on the device
            idx = [i for i in range(B) if psf.check_domain(sigmas[i])]
            out = [False] * B
            if idx:
                u = psf.ctx.hash_to_mat_zq_sha256([messages[i] for i in idx], psf.gp.n, 1, psf.gp.q)
                t = psf.f_a_batch_dev(pk, [sigmas[i] for i in idx])
                for i, v in zip(idx, (t == u).all(dim=1).cpu().tolist()):
                    out[i] = bool(v)
            return out
        good_shape = [isinstance(s, MatPolyOverZ) and s.get_num_rows() == psf.mbar and s.n == psf.gp.n for s in sigmas]
        idx = [i for i in range(B) if good_shape[i]]
        out = [False] * B
        if not idx:
            return out
        u = self.hash.hash_batch_dev(psf.ctx, [messages[i] for i in idx])   # SHA-256 on the device
        key = psf.ring.key(np.asarray(pk, np.uint32).reshape(psf.mbar, psf.gp.n))
        ok = psf.ring.fdh_verify(key, psf._sigma_tensor([sigmas[i] for i in idx]), u, psf.norm_bound())
        for i, v in zip(idx, psf.ctx.to_host_u8(ok)):
            out[i] = bool(v)
        return out
