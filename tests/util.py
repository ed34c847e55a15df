"""shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import hashlib

import numpy as np


def limbs(hexlist):
    return np.array([int(h, 16) for h in hexlist], dtype=np.uint64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def noncanonical(coeffs, fr_modulus):
    """+r on every third element: inputs in [0, 2r) as the prover produces."""
    out = coeffs.copy()
    mod = np.array([(fr_modulus >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    idx = np.arange(0, out.shape[0], 3)
    carry = np.zeros(idx.shape[0], dtype=np.uint64)
    for l in range(4):
        a = out[idx, l]
        s = a + mod[l]
        c1 = (s < a).astype(np.uint64)
        s2 = s + carry
        c2 = (s2 < s).astype(np.uint64)
        out[idx, l] = s2
        carry = c1 + c2
    return out


SCALAR_SEED = 0x9E3779B97F4A7C15
SRS_SEED = 0x5EED0F5EC2E7C0DE
NTT_SEED = 0x0123456789ABCDEF
CONST_SEED = 0x00C0FFEE00C0FFEE
