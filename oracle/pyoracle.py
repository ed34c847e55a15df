"""ctypes bindings for the TEST-ONLY checkers under oracle/.

* ``Oracle``  -> oracle/_build/liboracle.so  (oracle/bn254_oracle.c, our CPU restatement)
* ``Ref``     -> oracle/_ref/libbbref*.so     (the reference's own sources, built by oracle/Makefile)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under barretenberg_amd/ does.  Arrays are numpy uint64, little-endian limbs:
field element (4,), affine point (8,), Jacobian point (12,).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
u64p = C.POINTER(C.c_uint64)

FQ, FR = 0, 1
FFT, IFFT, COSET_FFT, COSET_IFFT, FFT_WITH_CONSTANT, IFFT_WITH_CONSTANT, COSET_FFT_WITH_CONSTANT = range(7)
NTT_KINDS = {
    "fft": FFT,
    "ifft": IFFT,
    "coset_fft": COSET_FFT,
    "coset_ifft": COSET_IFFT,
    "fft_with_constant": FFT_WITH_CONSTANT,
    "ifft_with_constant": IFFT_WITH_CONSTANT,
    "coset_fft_with_constant": COSET_FFT_WITH_CONSTANT,
}

FQ_MODULUS = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def aligned_empty(shape, dtype=np.uint64, align=64):
    """numpy array whose data pointer is `align`-byte aligned (the reference declares alignas(32))."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    raw = np.empty(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def aligned_copy(a, align=64):
    a = np.asarray(a, dtype=np.uint64)
    out = aligned_empty(a.shape, np.uint64, align)
    out[...] = a
    return out


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def to_int(limbs):
    limbs = np.asarray(limbs, dtype=np.uint64).reshape(-1)
    return sum(int(v) << (64 * i) for i, v in enumerate(limbs))


def from_int(v, nlimbs=4):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs)], dtype=np.uint64)


def build(force=False):
    """Compile oracle/_build/liboracle.so and (when /root/reference exists) oracle/_ref/*.so."""
    need = force or not os.path.exists(os.path.join(HERE, "_build", "liboracle.so"))
    if os.path.isdir("/root/reference/src/barretenberg"):
        need = need or not os.path.exists(os.path.join(HERE, "_ref", "libbbref.so"))
        need = need or not os.path.exists(os.path.join(HERE, "_ref", "libbbref_portable.so"))
    if need:
        subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL)


class Oracle:
    """oracle/bn254_oracle.c"""

    def __init__(self):
        path = os.path.join(HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.orc_get_optimal_bucket_width.restype = C.c_size_t
        L.orc_get_optimal_bucket_width.argtypes = [C.c_size_t]
        L.orc_get_wnaf_bits.restype = C.c_uint32
        L.orc_ntt.restype = C.c_int
        L.orc_batched_msm.restype = C.c_int
        L.orc_g1_on_curve_affine.restype = C.c_int
        L.orc_splitmix64.restype = C.c_uint64

    # ---- field ----
    def _bin(self, fn, f, a, b):
        r = np.zeros(4, dtype=np.uint64)
        fn(C.c_int(f), ptr(np.ascontiguousarray(a, dtype=np.uint64)), ptr(np.ascontiguousarray(b, dtype=np.uint64)), ptr(r))
        return r

    def _un(self, fn, f, a):
        r = np.zeros(4, dtype=np.uint64)
        fn(C.c_int(f), ptr(np.ascontiguousarray(a, dtype=np.uint64)), ptr(r))
        return r

    def mul(self, f, a, b): return self._bin(self.lib.orc_mul, f, a, b)
    def mul_coarse(self, f, a, b): return self._bin(self.lib.orc_mul_coarse, f, a, b)
    def add(self, f, a, b): return self._bin(self.lib.orc_add, f, a, b)
    def add_coarse(self, f, a, b): return self._bin(self.lib.orc_add_coarse, f, a, b)
    def sub(self, f, a, b): return self._bin(self.lib.orc_sub, f, a, b)
    def sub_coarse(self, f, a, b): return self._bin(self.lib.orc_sub_coarse, f, a, b)
    def sqr(self, f, a): return self._un(self.lib.orc_sqr, f, a)
    def sqr_coarse(self, f, a): return self._un(self.lib.orc_sqr_coarse, f, a)
    def reduce_once(self, f, a): return self._un(self.lib.orc_reduce_once, f, a)
    def neg(self, f, a): return self._un(self.lib.orc_neg, f, a)
    def to_mont(self, f, a): return self._un(self.lib.orc_to_mont, f, a)
    def from_mont(self, f, a): return self._un(self.lib.orc_from_mont, f, a)
    def invert(self, f, a): return self._un(self.lib.orc_invert, f, a)

    def const(self, f, name):
        self.lib.orc_const.restype = u64p
        p = self.lib.orc_const(C.c_int(f), name.encode())
        return np.array([p[i] for i in range(4)], dtype=np.uint64)

    def root_of_unity(self, log2n):
        r = np.zeros(4, dtype=np.uint64)
        self.lib.orc_get_root_of_unity(C.c_size_t(log2n), ptr(r))
        return r

    # ---- scalars ----
    def split_endo(self, k):
        k1 = np.zeros(2, dtype=np.uint64)
        k2 = np.zeros(2, dtype=np.uint64)
        self.lib.orc_split_endo(ptr(np.ascontiguousarray(k, dtype=np.uint64)), ptr(k1), ptr(k2))
        return k1, k2

    def fixed_wnaf(self, scalar2, wnaf_bits):
        entries = (127 + wnaf_bits - 1) // wnaf_bits
        w = np.zeros(entries, dtype=np.uint32)
        skew = C.c_uint8(0)
        self.lib.orc_fixed_wnaf(ptr(np.ascontiguousarray(scalar2, dtype=np.uint64)), w.ctypes.data_as(C.POINTER(C.c_uint32)),
                                C.byref(skew), C.c_size_t(1), C.c_size_t(wnaf_bits))
        return w, int(skew.value)

    def random_scalars(self, seed, n):
        out = aligned_empty((n, 4))
        self.lib.orc_random_scalars(C.c_uint64(seed), C.c_size_t(n), ptr(out))
        return out

    # ---- group ----
    def g1_dbl(self, p):
        r = np.zeros(12, dtype=np.uint64)
        self.lib.orc_g1_dbl(ptr(np.ascontiguousarray(p, dtype=np.uint64)), ptr(r))
        return r

    def g1_mixed_add(self, p, q):
        r = np.zeros(12, dtype=np.uint64)
        self.lib.orc_g1_mixed_add(ptr(np.ascontiguousarray(p, dtype=np.uint64)), ptr(np.ascontiguousarray(q, dtype=np.uint64)), ptr(r))
        return r

    def g1_add(self, p, q):
        r = np.zeros(12, dtype=np.uint64)
        self.lib.orc_g1_add(ptr(np.ascontiguousarray(p, dtype=np.uint64)), ptr(np.ascontiguousarray(q, dtype=np.uint64)), ptr(r))
        return r

    def g1_normalize(self, p):
        r = np.zeros(12, dtype=np.uint64)
        self.lib.orc_g1_normalize(ptr(np.ascontiguousarray(p, dtype=np.uint64)), ptr(r))
        return r

    def g1_one_affine(self):
        r = np.zeros(8, dtype=np.uint64)
        self.lib.orc_g1_one_affine(ptr(r))
        return r

    def g1_scalar_mul(self, p_affine, scalar_mont):
        r = np.zeros(12, dtype=np.uint64)
        self.lib.orc_g1_scalar_mul(ptr(np.ascontiguousarray(p_affine, dtype=np.uint64)),
                                   ptr(np.ascontiguousarray(scalar_mont, dtype=np.uint64)), ptr(r))
        return r

    def g1_on_curve(self, p_affine):
        return bool(self.lib.orc_g1_on_curve_affine(ptr(np.ascontiguousarray(p_affine, dtype=np.uint64))))

    @staticmethod
    def is_infinity(p):
        return bool((int(p[7]) >> 63) & 1)

    # ---- MSM ----
    def optimal_bucket_width(self, n):
        return int(self.lib.orc_get_optimal_bucket_width(n))

    def make_srs(self, x_mont, n):
        out = aligned_empty((n, 8))
        self.lib.orc_make_srs(ptr(np.ascontiguousarray(x_mont, dtype=np.uint64)), C.c_size_t(n), ptr(out))
        return out

    def point_table(self, points):
        n = points.shape[0]
        table = aligned_empty((2 * n, 8))
        self.lib.orc_generate_point_table(ptr(np.ascontiguousarray(points)), ptr(table), C.c_size_t(n))
        return table

    def pippenger(self, scalars, table, n=None, forced_bucket_width=0):
        n = scalars.shape[0] if n is None else n
        out = np.zeros(12, dtype=np.uint64)
        self.lib.orc_pippenger(ptr(scalars), ptr(table), C.c_size_t(n), C.c_size_t(forced_bucket_width), ptr(out))
        return out

    def msm_affine(self, scalars, table, n=None, forced_bucket_width=0):
        """normalised result: (x, y, z=one) or infinity flag, as batched_scalar_multiplications returns it"""
        return self.g1_normalize_or_inf(self.pippenger(scalars, table, n, forced_bucket_width))

    def g1_normalize_or_inf(self, p):
        if self.is_infinity(p):
            r = np.zeros(12, dtype=np.uint64)
            r[7] = np.uint64(1 << 63)
            return r
        return self.g1_normalize(p)

    # ---- NTT ----
    def ntt(self, coeffs, kind, constant=None):
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        out = aligned_copy(coeffs)
        n = out.shape[0]
        cp = ptr(np.ascontiguousarray(constant, dtype=np.uint64)) if constant is not None else None
        rc = self.lib.orc_ntt(ptr(out), C.c_size_t(n), C.c_int(kind), cp)
        if rc:
            raise ValueError("orc_ntt rc=%d" % rc)
        return out

    def evaluate(self, coeffs, z):
        r = np.zeros(4, dtype=np.uint64)
        self.lib.orc_evaluate(ptr(np.ascontiguousarray(coeffs)), ptr(np.ascontiguousarray(z, dtype=np.uint64)),
                              C.c_size_t(coeffs.shape[0]), ptr(r))
        return r


class Ref:
    """The reference's own compiled sources (oracle/_ref).  asm=True -> x86-64 asm path."""

    def __init__(self, asm=True):
        name = "libbbref.so" if asm else "libbbref_portable.so"
        path = os.path.join(HERE, "_ref", name)
        if not os.path.exists(path):
            build()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        if asm and not self.cpu_has_bmi2_adx():
            raise RuntimeError("host CPU lacks BMI2/ADX needed by the reference asm path")
        L = self.lib = C.CDLL(path)
        L.ref_get_optimal_bucket_width.restype = C.c_size_t
        L.ref_get_optimal_bucket_width.argtypes = [C.c_size_t]
        L.ref_ntt.restype = C.c_int
        L.ref_max_threads.restype = C.c_int
        L.ref_uses_asm.restype = C.c_int

    @staticmethod
    def cpu_has_bmi2_adx():
        try:
            with open("/proc/cpuinfo") as fh:
                for line in fh:
                    if line.startswith("flags"):
                        flags = set(line.split(":", 1)[1].split())
                        return "bmi2" in flags and "adx" in flags
        except OSError:
            pass
        return False

    @staticmethod
    def available(asm=True):
        name = "libbbref.so" if asm else "libbbref_portable.so"
        return os.path.exists(os.path.join(HERE, "_ref", name)) and (not asm or Ref.cpu_has_bmi2_adx())

    def max_threads(self):
        return int(self.lib.ref_max_threads())

    def set_threads(self, n):
        self.lib.ref_set_threads(C.c_int(n))

    OPS = {"mul": 0, "sqr": 1, "add": 2, "sub": 3, "mul_coarse": 4, "add_coarse": 5, "sub_coarse": 6, "reduce_once": 7,
           "to_mont": 8, "from_mont": 9, "invert": 10, "neg": 11, "sqr_coarse": 12}

    def field_op(self, f, op, a, b=None):
        r = aligned_empty((4,))
        a = aligned_copy(a)
        bp = ptr(aligned_copy(b)) if b is not None else None
        self.lib.ref_field_op(C.c_int(f), C.c_int(self.OPS[op]), ptr(a), bp, ptr(r))
        return np.array(r)

    def split_endo(self, k):
        k1 = np.zeros(2, dtype=np.uint64)
        k2 = np.zeros(2, dtype=np.uint64)
        self.lib.ref_split_endo(ptr(aligned_copy(k)), ptr(k1), ptr(k2))
        return k1, k2

    def fixed_wnaf(self, scalar2, wnaf_bits):
        entries = (127 + wnaf_bits - 1) // wnaf_bits
        w = np.zeros(entries, dtype=np.uint32)
        skew = C.c_uint8(0)
        self.lib.ref_fixed_wnaf(ptr(np.ascontiguousarray(scalar2, dtype=np.uint64)), w.ctypes.data_as(C.POINTER(C.c_uint32)),
                                C.byref(skew), C.c_size_t(1), C.c_size_t(wnaf_bits))
        return w, int(skew.value)

    def g1_op(self, op, p1, p2=None):
        code = {"dbl": 0, "mixed_add": 1, "add": 2, "normalize": 3}[op]
        r = aligned_empty((12,))
        p2p = ptr(aligned_copy(p2)) if p2 is not None else None
        self.lib.ref_g1_op(C.c_int(code), ptr(aligned_copy(p1)), p2p, ptr(r))
        return np.array(r)

    def g1_scalar_mul(self, p_affine, scalar_mont):
        r = aligned_empty((12,))
        self.lib.ref_g1_scalar_mul(ptr(aligned_copy(p_affine)), ptr(aligned_copy(scalar_mont)), ptr(r))
        return np.array(r)

    def optimal_bucket_width(self, n):
        return int(self.lib.ref_get_optimal_bucket_width(n))

    def point_table(self, points):
        n = points.shape[0]
        table = aligned_empty((2 * n, 8))
        table[:n] = points
        self.lib.ref_generate_point_table(ptr(table), ptr(table), C.c_size_t(n))
        return table

    def pippenger(self, scalars, table, n=None, forced_bucket_width=0):
        n = scalars.shape[0] if n is None else n
        out = aligned_empty((12,))
        assert scalars.ctypes.data % 32 == 0 and table.ctypes.data % 32 == 0
        self.lib.ref_pippenger(ptr(scalars), ptr(table), C.c_size_t(n), C.c_size_t(forced_bucket_width), ptr(out))
        return np.array(out)

    def pippenger_low_memory(self, scalars, points, n=None):
        """the reference's pippenger_low_memory: `points` is the plain n-entry table; `scalars` are clobbered"""
        n = scalars.shape[0] if n is None else n
        out = aligned_empty((12,))
        assert scalars.ctypes.data % 32 == 0 and points.ctypes.data % 32 == 0
        self.lib.ref_pippenger_low_memory(ptr(scalars), ptr(points), C.c_size_t(n), ptr(out))
        return np.array(out)

    def batched_msm(self, scalars_list, tables_list):
        num = len(scalars_list)
        n = scalars_list[0].shape[0]
        sp = (u64p * num)(*[ptr(s) for s in scalars_list])
        tp = (u64p * num)(*[ptr(t) for t in tables_list])
        outs = aligned_empty((num, 12))
        self.lib.ref_batched_msm(sp, tp, C.c_size_t(n), C.c_size_t(num), ptr(outs))
        return np.array(outs)

    def prepare_domain(self, n):
        self.lib.ref_prepare_domain(C.c_size_t(n))

    def ntt_inplace(self, coeffs, kind, constant=None):
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        assert coeffs.ctypes.data % 32 == 0
        cp = ptr(aligned_copy(constant)) if constant is not None else None
        rc = self.lib.ref_ntt(ptr(coeffs), C.c_size_t(coeffs.shape[0]), C.c_int(kind), cp)
        if rc:
            raise ValueError("ref_ntt rc=%d" % rc)
        return coeffs

    def ntt(self, coeffs, kind, constant=None):
        return np.array(self.ntt_inplace(aligned_copy(coeffs), kind, constant))

    def evaluate(self, coeffs, z):
        r = aligned_empty((4,))
        self.lib.ref_evaluate(ptr(aligned_copy(coeffs)), ptr(aligned_copy(z)), C.c_size_t(coeffs.shape[0]), ptr(r))
        return np.array(r)

    # ---- the O(n) helpers between the transforms ----
    def batch_invert(self, v):
        out = aligned_copy(v)
        self.lib.ref_batch_invert(ptr(out), C.c_size_t(out.shape[0]))
        return np.array(out)

    def kate_opening(self, src, z):
        """-> (dest, F(z)); dest is left coarse by the reference (polynomial_arithmetic.cpp:580-588)"""
        src = aligned_copy(src)
        dest = aligned_empty(src.shape)
        f = aligned_empty((4,))
        self.lib.ref_kate_opening(ptr(src), ptr(dest), ptr(aligned_copy(z)), C.c_size_t(src.shape[0]), ptr(f))
        return np.array(dest), np.array(f)

    def lagrange_l1_fft(self, n_src, n_target):
        out = aligned_empty((n_target + 8, 4))
        self.lib.ref_lagrange_l1_fft(ptr(out), C.c_size_t(n_src), C.c_size_t(n_target))
        return np.array(out[:n_target])

    def divide_by_pseudo_vanishing(self, coeffs, n_src, n_target):
        out = aligned_copy(coeffs)
        self.lib.ref_divide_by_pseudo_vanishing(ptr(out), C.c_size_t(n_src), C.c_size_t(n_target))
        return np.array(out)

    def pointwise_mul(self, a, b):
        out = aligned_empty(a.shape)
        self.lib.ref_pointwise_mul(ptr(aligned_copy(a)), ptr(aligned_copy(b)), ptr(out), C.c_size_t(a.shape[0]))
        return np.array(out)

    def lagrange_evaluations(self, z, n):
        out = aligned_empty((3, 4))
        self.lib.ref_lagrange_evaluations(ptr(aligned_copy(z)), C.c_size_t(n), ptr(out))
        return np.array(out)


class PolyOracle:
    """Pure-Python (big integer) restatement of the reference's O(n) polynomial helpers, for small cases only.
    Arrays in / out are (n, 4) uint64 Montgomery limbs like everywhere else; results are canonical.  Each method cites the
    reference loop it follows; tests/test_oracle.py pins them against the reference build (Ref above)."""
    R = FR_MODULUS
    MONT = (1 << 256) % FR_MODULUS
    MONT_INV = pow((1 << 256) % FR_MODULUS, -1, FR_MODULUS)
    # fr.hpp:59-63: the 2^28-th root of unity, read out of its Montgomery form
    ROOT28 = 0x1860EF942963F9E756452AC01EB203D8A22BF3742445FFD6636E735580D13D9C * pow(1 << 256, -1, FR_MODULUS) % FR_MODULUS
    GENERATOR, K2 = 5, 7  # fr.hpp:65-79 (multiplicative_generator, alternate_multiplicative_generator)

    @classmethod
    def plain(cls, a):
        a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
        return [to_int(row) * cls.MONT_INV % cls.R for row in a]

    @classmethod
    def mont(cls, values):
        out = np.zeros((len(values), 4), dtype=np.uint64)
        for i, v in enumerate(values):
            out[i] = from_int(v % cls.R * cls.MONT % cls.R)
        return out

    @classmethod
    def root(cls, log2n):  # field.hpp:487-494
        return pow(cls.ROOT28, 1 << (28 - log2n), cls.R)

    @classmethod
    def evaluate(cls, coeffs, z):  # polynomial_arithmetic.cpp:337-373
        zz, acc, zp = cls.plain(z)[0], 0, 1
        for c in cls.plain(coeffs):
            acc = (acc + c * zp) % cls.R
            zp = zp * zz % cls.R
        return cls.mont([acc])[0]

    @classmethod
    def batch_invert(cls, v):  # fields/field.hpp:503-522 (Montgomery's trick; same values as element-wise inversion)
        return cls.mont([pow(x, -1, cls.R) for x in cls.plain(v)])

    @classmethod
    def product_scan(cls, v, reverse=False, inclusive=False):  # prover.cpp:194-202 is the exclusive prefix form
        x = cls.plain(v)
        if reverse:
            x = x[::-1]
        out, acc = [], 1
        for a in x:
            if inclusive:
                acc = acc * a % cls.R
                out.append(acc)
            else:
                out.append(acc)
                acc = acc * a % cls.R
        return cls.mont(out[::-1] if reverse else out)

    @classmethod
    def kate_opening(cls, src, z):  # polynomial_arithmetic.cpp:562-591, the reference's bottom-up recurrence
        f, zz = cls.plain(src), cls.plain(z)[0]
        fz = 0
        for c in reversed(f):
            fz = (fz * zz + c) % cls.R
        divisor = pow(-zz % cls.R, -1, cls.R)
        dest = [(f[0] - fz) * divisor % cls.R]
        for i in range(1, len(f)):
            dest.append((f[i] - dest[i - 1]) * divisor % cls.R)
        return cls.mont(dest), cls.mont([fz])[0]

    @classmethod
    def lagrange_l1_fft(cls, n_src, n_target):  # polynomial_arithmetic.cpp:381-476
        ls, lt = n_src.bit_length() - 1, n_target.bit_length() - 1
        k = n_target // n_src
        wt, gn, wk = cls.root(lt), pow(cls.GENERATOR, n_src, cls.R), cls.root(lt - ls)
        numer = [(gn * pow(wk, j, cls.R) - 1) * pow(n_src, -1, cls.R) % cls.R for j in range(k)]
        out, x = [], cls.GENERATOR
        for i in range(n_target):
            out.append(pow(x - 1, -1, cls.R) * numer[i % k] % cls.R)
            x = x * wt % cls.R
        return cls.mont(out)

    @classmethod
    def divide_by_pseudo_vanishing(cls, coeffs, n_src, n_target):  # polynomial_arithmetic.cpp:478-560
        ls, lt = n_src.bit_length() - 1, n_target.bit_length() - 1
        k = n_target // n_src
        wt, gn, wk = cls.root(lt), pow(cls.GENERATOR, n_src, cls.R), cls.root(lt - ls)
        inv = [pow(gn * pow(wk, j, cls.R) - 1, -1, cls.R) for j in range(k)]
        last = pow(cls.root(ls), -1, cls.R)  # w^(n-1)
        out, x = [], cls.GENERATOR
        for i, c in enumerate(cls.plain(coeffs)):
            out.append(c * inv[i % k] % cls.R * ((x - last) % cls.R) % cls.R)
            x = x * wt % cls.R
        return cls.mont(out)

    @classmethod
    def permutation_lagrange_base(cls, mapping, n):  # waffle/proof_system/permutation.hpp:15-87
        w = cls.root(n.bit_length() - 1)
        out = []
        for m in mapping:
            m = int(m)
            v = pow(w, m & ((1 << 29) - 1), cls.R)
            t = (m >> 30) & 3
            out.append(v * (cls.K2 if t == 2 else cls.GENERATOR if t == 1 else 1) % cls.R)
        return cls.mont(out)

    @classmethod
    def pointwise_mul(cls, a, b):  # polynomial_arithmetic.cpp:328-335
        return cls.mont([x * y % cls.R for x, y in zip(cls.plain(a), cls.plain(b))])
