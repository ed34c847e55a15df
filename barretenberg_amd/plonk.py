"""Host-side mirror of the reference's circuit front end for the resident PLONK prover (bbgpu_plonk_*).

``StandardComposer`` follows waffle::StandardComposer / ComposerBase
(src/barretenberg/waffle/composer/standard_composer.cpp:13-220, composer_base.hpp:131-199) and ``field_t`` the witness
arithmetic of plonk::stdlib::field_t (src/barretenberg/waffle/stdlib/field/field.tcc:124-252): the same gates, wire
indices, selector values and sigma mappings, so ``preprocess()`` yields exactly the state the reference's composer hands
its Prover.  Values are plain Python integers mod r here and converted to the reference's memory format (4 x u64 limbs,
Montgomery 2^256) at the end.  ``Prover`` wraps the C ABI (include/bbgpu.h); proving itself runs on the GPU only.
"""
import ctypes as C

import numpy as np

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
NO_WITNESS = 0xFFFFFFFF
LEFT, RIGHT, OUTPUT = 0, 1 << 30, 1 << 31  # ComposerBase::WireType, composer_base.hpp:73-79

PROOF_POINTS = ["W_L", "W_R", "W_O", "Z_1", "T_LO", "T_MID", "T_HI", "PI_Z", "PI_Z_OMEGA"]
VK_POINTS = ["SIGMA_1", "SIGMA_2", "SIGMA_3", "Q_M", "Q_L", "Q_R", "Q_O", "Q_C"]
VK_POINTS_BOOL = VK_POINTS + ["Q_BL", "Q_BR", "Q_BO"]
PROOF_EVALS = ["w_l_eval", "w_r_eval", "w_o_eval", "sigma_1_eval", "sigma_2_eval", "z_1_shifted_eval", "linear_eval"]


def to_montgomery_limbs(values):
    """plain integers -> (len, 4) uint64 array of x * 2^256 mod r"""
    out = np.empty((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        m = (v << 256) % FR_MODULUS
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


class StandardComposer:
    def __init__(self):
        self.n = 0
        self.variables = []
        self.wire_epicycles = []
        self.w_l, self.w_r, self.w_o = [], [], []
        self.q_m, self.q_l, self.q_r, self.q_o, self.q_c = [], [], [], [], []

    # composer_base.hpp:131-136
    def add_variable(self, value):
        self.variables.append(value % FR_MODULUS)
        self.wire_epicycles.append([])
        return len(self.variables) - 1

    def get_variable(self, index):
        return self.variables[index]

    def _gate(self, a, b, c, q_m, q_l, q_r, q_o, q_c):
        self.w_l.append(a); self.w_r.append(b); self.w_o.append(c)
        self.q_m.append(q_m % FR_MODULUS); self.q_l.append(q_l % FR_MODULUS); self.q_r.append(q_r % FR_MODULUS)
        self.q_o.append(q_o % FR_MODULUS); self.q_c.append(q_c % FR_MODULUS)
        self.wire_epicycles[a].append((self.n, LEFT))
        self.wire_epicycles[b].append((self.n, RIGHT))
        self.wire_epicycles[c].append((self.n, OUTPUT))
        self.n += 1

    # standard_composer.cpp:13-36
    def create_add_gate(self, a, b, c, a_scaling, b_scaling, c_scaling, const_scaling):
        self._gate(a, b, c, 0, a_scaling, b_scaling, c_scaling, const_scaling)

    # standard_composer.cpp:38-63
    def create_mul_gate(self, a, b, c, mul_scaling, c_scaling, const_scaling):
        self._gate(a, b, c, mul_scaling, 0, 0, c_scaling, const_scaling)

    # standard_composer.cpp:65-89
    def create_bool_gate(self, a):
        self._gate(a, a, a, 1, 0, 0, -1, 0)

    # standard_composer.cpp:91-116
    def create_poly_gate(self, a, b, c, q_m, q_l, q_r, q_o, q_c):
        self._gate(a, b, c, q_m, q_l, q_r, q_o, q_c)

    # standard_composer.cpp:163-220 + composer_base.hpp:164-199 (compute_sigma_permutations)
    def preprocess(self):
        n = self.n
        log2_n = (n + 1).bit_length() - 1
        if (1 << log2_n) != n + 1:
            log2_n += 1
        new_n = 1 << log2_n
        variables = self.variables + [0]
        zero_idx = len(variables) - 1
        pad = new_n - n
        w_l = self.w_l + [zero_idx] * pad
        w_r = self.w_r + [zero_idx] * pad
        w_o = self.w_o + [zero_idx] * pad
        sel = [q + [0] * pad for q in (self.q_m, self.q_l, self.q_r, self.q_o, self.q_c)]
        sigma = [np.arange(new_n, dtype=np.uint32) + np.uint32(t) for t in (LEFT, RIGHT, OUTPUT)]
        for cyc in self.wire_epicycles:
            for j, (gate, wire) in enumerate(cyc):
                nxt_gate, nxt_wire = cyc[0] if j == len(cyc) - 1 else cyc[j + 1]
                sigma[wire >> 30][gate] = np.uint32((nxt_gate + nxt_wire) & 0xFFFFFFFF)
        return {
            "n": new_n,
            "w_l": to_montgomery_limbs([variables[i] for i in w_l]),
            "w_r": to_montgomery_limbs([variables[i] for i in w_r]),
            "w_o": to_montgomery_limbs([variables[i] for i in w_o]),
            "sigma_1_mapping": sigma[0], "sigma_2_mapping": sigma[1], "sigma_3_mapping": sigma[2],
            "q_m": to_montgomery_limbs(sel[0]), "q_l": to_montgomery_limbs(sel[1]), "q_r": to_montgomery_limbs(sel[2]),
            "q_o": to_montgomery_limbs(sel[3]), "q_c": to_montgomery_limbs(sel[4]),
        }


class BoolComposer(StandardComposer):
    """waffle::BoolComposer (composer/bool_composer.hpp:8-46, bool_composer.cpp:13-143): boolean constraints are not gates of their
    own but selectors q_bl / q_br / q_bo on the wires of the existing gates, checked by the bool widget next to the arithmetic one"""

    def __init__(self):
        super().__init__()
        self.is_bool = []
        self.zero_idx = self.add_variable(0)  # bool_composer.hpp:16

    def add_variable(self, value):
        self.is_bool.append(False)
        return super().add_variable(value)

    # bool_composer.cpp:23-29
    def create_bool_gate(self, a):
        self.is_bool[a] = True

    # bool_composer.cpp:68-143 (process_bool_gates + preprocess)
    def preprocess(self):
        n = self.n
        log2_n = (n + 1).bit_length() - 1
        if (1 << log2_n) != n + 1:
            log2_n += 1
        new_n = 1 << log2_n
        pad = new_n - n
        qb = [[1 if self.is_bool[w[i]] else 0 for i in range(n)] + [0] * pad for w in (self.w_l, self.w_r, self.w_o)]
        w_l = self.w_l + [self.zero_idx] * pad
        w_r = self.w_r + [self.zero_idx] * pad
        w_o = self.w_o + [self.zero_idx] * pad
        sel = [q + [0] * pad for q in (self.q_m, self.q_l, self.q_r, self.q_o, self.q_c)]
        sigma = [np.arange(new_n, dtype=np.uint32) + np.uint32(t) for t in (LEFT, RIGHT, OUTPUT)]
        for cyc in self.wire_epicycles:
            for j, (gate, wire) in enumerate(cyc):
                nxt_gate, nxt_wire = cyc[0] if j == len(cyc) - 1 else cyc[j + 1]
                sigma[wire >> 30][gate] = np.uint32((nxt_gate + nxt_wire) & 0xFFFFFFFF)
        v = self.variables
        return {
            "n": new_n,
            "w_l": to_montgomery_limbs([v[i] for i in w_l]), "w_r": to_montgomery_limbs([v[i] for i in w_r]),
            "w_o": to_montgomery_limbs([v[i] for i in w_o]),
            "sigma_1_mapping": sigma[0], "sigma_2_mapping": sigma[1], "sigma_3_mapping": sigma[2],
            "q_m": to_montgomery_limbs(sel[0]), "q_l": to_montgomery_limbs(sel[1]), "q_r": to_montgomery_limbs(sel[2]),
            "q_o": to_montgomery_limbs(sel[3]), "q_c": to_montgomery_limbs(sel[4]),
            "q_bl": to_montgomery_limbs(qb[0]), "q_br": to_montgomery_limbs(qb[1]), "q_bo": to_montgomery_limbs(qb[2]),
        }


def bool_circuit(num_gates):
    """the BoolComposer fixture circuit of oracle/plonk_driver.cpp (BB_CIRCUIT=bool): num_gates / 2 pairs of bits a, b constrained
    boolean, c = a b (mul gate), d = a + c (add gate)"""
    composer = BoolComposer()
    for i in range(num_gates // 2):
        abit, bbit = (i * 7 + 1) & 1, ((i * 5 + 3) >> 1) & 1
        a, b = composer.add_variable(abit), composer.add_variable(bbit)
        c, d = composer.add_variable(abit & bbit), composer.add_variable(abit + (abit & bbit))
        composer.create_bool_gate(a)
        composer.create_bool_gate(b)
        composer.create_mul_gate(a, b, c, 1, -1, 0)
        composer.create_add_gate(a, c, d, 1, 1, -1, 0)
    return composer


class field_t:
    """plonk::stdlib::field_t: value = multiplicative_constant * witness + additive_constant (field.tcc:11-37)"""

    def __init__(self, composer, witness_index=NO_WITNESS, additive_constant=0, multiplicative_constant=1):
        self.context = composer
        self.witness_index = witness_index
        self.additive_constant = additive_constant % FR_MODULUS
        self.multiplicative_constant = multiplicative_constant % FR_MODULUS

    @staticmethod
    def witness(composer, value):
        return field_t(composer, composer.add_variable(value))

    @staticmethod
    def constant(composer, value):
        return field_t(composer, NO_WITNESS, value, 0)

    # field.tcc:124-181
    def __add__(self, other):
        ctx = self.context
        if self.witness_index == other.witness_index:
            return field_t(ctx, self.witness_index, self.additive_constant + other.additive_constant,
                           self.multiplicative_constant + other.multiplicative_constant)
        if self.witness_index != NO_WITNESS and other.witness_index == NO_WITNESS:
            return field_t(ctx, self.witness_index, self.additive_constant + other.additive_constant, self.multiplicative_constant)
        if self.witness_index == NO_WITNESS and other.witness_index != NO_WITNESS:
            return field_t(ctx, other.witness_index, self.additive_constant + other.additive_constant, other.multiplicative_constant)
        left, right = ctx.get_variable(self.witness_index), ctx.get_variable(other.witness_index)
        out = (left * self.multiplicative_constant + right * other.multiplicative_constant + self.additive_constant + other.additive_constant) % FR_MODULUS
        res = field_t(ctx, ctx.add_variable(out))
        ctx.create_add_gate(self.witness_index, other.witness_index, res.witness_index, self.multiplicative_constant,
                            other.multiplicative_constant, -1, self.additive_constant + other.additive_constant)
        return res

    # field.tcc:192-252
    def __mul__(self, other):
        ctx = self.context
        if self.witness_index == NO_WITNESS and other.witness_index == NO_WITNESS:
            return field_t(ctx, NO_WITNESS, self.additive_constant * other.additive_constant, 1)
        if self.witness_index != NO_WITNESS and other.witness_index == NO_WITNESS:
            return field_t(ctx, self.witness_index, self.additive_constant * other.additive_constant,
                           self.multiplicative_constant * other.additive_constant)
        if self.witness_index == NO_WITNESS and other.witness_index != NO_WITNESS:
            return field_t(ctx, other.witness_index, self.additive_constant * other.additive_constant,
                           other.multiplicative_constant * self.additive_constant)
        q_c = self.additive_constant * other.additive_constant % FR_MODULUS
        q_r = self.additive_constant * other.multiplicative_constant % FR_MODULUS
        q_l = self.multiplicative_constant * other.additive_constant % FR_MODULUS
        q_m = self.multiplicative_constant * other.multiplicative_constant % FR_MODULUS
        left, right = ctx.get_variable(self.witness_index), ctx.get_variable(other.witness_index)
        out = (left * right * q_m + left * q_l + right * q_r + q_c) % FR_MODULUS
        res = field_t(ctx, ctx.add_variable(out))
        ctx.create_poly_gate(self.witness_index, other.witness_index, res.witness_index, q_m, q_l, q_r, -1, q_c)
        return res


def bench_circuit(num_gates, a0, b0):
    """the add/mul chain of the reference's PLONK benchmark (src/barretenberg/test/benchmarks/bench_plonk.cpp:25-37) with given
    plain witness values; returns the composer"""
    composer = StandardComposer()
    a, b = field_t.witness(composer, a0), field_t.witness(composer, b0)
    for _ in range(num_gates // 4 - 4):
        c = a + b
        c = a * c
        a = b * b
        b = c * c
    return composer


class _Circuit(C.Structure):
    _fields_ = [("n", C.c_size_t)] + [(k, C.c_void_p) for k in ("w_l", "w_r", "w_o", "sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping",
                                                               "q_m", "q_l", "q_r", "q_o", "q_c", "q_bl", "q_br", "q_bo")]


class Prover:
    """waffle::Prover over the C ABI: Prover(gpu, circuit_state, srs_handle).construct_proof()"""

    def __init__(self, gpu, state, srs_handle):
        self.gpu = gpu
        L = gpu.lib
        L.bbgpu_plonk_prover_create.argtypes = [C.POINTER(_Circuit), C.c_int]
        L.bbgpu_plonk_construct_proof.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.bbgpu_plonk_last_challenges.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        L.bbgpu_plonk_last_timing.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.bbgpu_plonk_prover_set_witness.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.n = int(state["n"])
        self._keep = {k: np.ascontiguousarray(state[k]) for k in state if k != "n"}
        c = _Circuit()
        c.n = self.n
        for k, a in self._keep.items():
            want = np.uint32 if k.endswith("mapping") else np.uint64
            assert a.dtype == want and a.shape[0] == self.n, k
            setattr(c, k, a.ctypes.data)
        self.handle = gpu._chk(L.bbgpu_plonk_prover_create(C.byref(c), srs_handle))

    def set_witness(self, w_l, w_r, w_o):
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in (w_l, w_r, w_o)]
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_prover_set_witness(self.handle, *[a.ctypes.data for a in arrs]))

    def construct_proof(self):
        """-> (100,) uint64: nine affine commitments then seven evaluations (waffle_types.hpp:18-45)"""
        out = np.zeros(100, dtype=np.uint64)
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_construct_proof(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def preprocess(self):
        """waffle::preprocess(prover): -> dict of the eight verification-key commitments, each (8,) uint64 affine"""
        out = np.zeros(88, dtype=np.uint64)
        self.gpu.lib.bbgpu_plonk_preprocess.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_preprocess(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return {k: out[8 * i:8 * i + 8] for i, k in enumerate(VK_POINTS_BOOL if "q_bl" in self._keep else VK_POINTS)}

    def challenges(self):
        out = np.zeros(20, dtype=np.uint64)
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_last_challenges(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return {k: out[4 * i:4 * i + 4] for i, k in enumerate(("beta", "gamma", "alpha", "z", "nu"))}

    def timing(self):
        buf = (C.c_double * 4)()
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_last_timing(self.handle, buf))
        return {"total_ms": buf[0], "commitments_ms": buf[1], "rest_ms": buf[2], "first_use_preparation_ms": buf[3]}

    def destroy(self):
        if self.handle is not None:
            self.gpu.lib.bbgpu_plonk_prover_destroy(self.handle)
            self.handle = None


def hex4(limbs):
    return "%016x%016x%016x%016x" % (int(limbs[3]), int(limbs[2]), int(limbs[1]), int(limbs[0]))


def proof_lines(n, proof):
    """the text form oracle/plonk_driver.cpp prints and tests/golden/plonk_proofs.json stores"""
    out = ["n %d" % n]
    for i, name in enumerate(PROOF_POINTS):
        out.append("%s.x %s" % (name, hex4(proof[8 * i:8 * i + 4])))
        out.append("%s.y %s" % (name, hex4(proof[8 * i + 4:8 * i + 8])))
    for i, name in enumerate(PROOF_EVALS):
        out.append("%s %s" % (name, hex4(proof[72 + 4 * i:76 + 4 * i])))
    return out


def proof_from_lines(lines):
    """inverse of proof_lines: -> (n, (100,) uint64)"""
    n = int(lines[0].split()[1])
    words = []
    for ln in lines[1:1 + 18 + 7]:
        h = ln.split()[1]
        words += [int(h[16 * (3 - k):16 * (4 - k)], 16) for k in range(4)]
    return n, np.array(words, dtype=np.uint64)
