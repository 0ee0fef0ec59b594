"""pk_encryption.rs:55-82 (PKEncryptionScheme) and :120-167 (GenericMultiBitEncryption)."""
from __future__ import annotations

import math
from fractions import Fraction

from ...math import MathError, as_fraction, is_prime  # noqa: F401


class PKEncryptionScheme:
    """gen() -> (pk, sk); enc(pk, message) -> cipher; dec(sk, cipher) -> int."""

    def gen(self):
        raise NotImplementedError

    def enc(self, pk, message: int):
        raise NotImplementedError

    def dec(self, sk, cipher) -> int:
        raise NotImplementedError


class GenericMultiBitEncryption(PKEncryptionScheme):
    """pk_encryption.rs:120-167: LSB-first bitwise encryption of |message|."""

    def enc_multiple_bits(self, pk, message: int):
        message = abs(int(message))
        bits = [(message >> i) & 1 for i in range(message.bit_length())]  # Z::to_bits, LSB first
        return [self.enc(pk, b) for b in bits]

    def dec_multiple_bits(self, sk, cipher) -> int:
        out = 0
        for i, c in enumerate(cipher):
            if self.dec(sk, c) != 0:
                out |= 1 << i
        return out


def sqrt_q(x) -> Fraction:
    """Rational approximation of sqrt used only inside parameter checks (qfall-math Z::sqrt -> Q)."""
    return Fraction(math.sqrt(float(x)))


def log2_q(x) -> Fraction:
    return Fraction(math.log2(float(x)))


def largest_prime_at_most(upper: int, lower: int) -> int:
    """Deterministic stand-in for Z::sample_prime_uniform(lower, upper): the largest
    prime in [lower, upper]; any prime of the interval is a valid reference parameter."""
    p = upper
    while p >= lower:
        if is_prime(p):
            return p
        p -= 1
    raise MathError(f"no prime in [{lower}, {upper}]")
