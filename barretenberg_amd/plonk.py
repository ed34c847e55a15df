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
VK_POINTS_MIMC = VK_POINTS + ["Q_MIMC_COEFFICIENT", "Q_MIMC_SELECTOR"]
VK_POINTS_EXTENDED = VK_POINTS + ["Q_O_NEXT", "Q_BL", "Q_BR", "Q_BO"]  # ExtendedComposer: arithmetic, sequential, bool widgets
PROOF_EVALS_WIDGET = ["w_l_shifted_eval", "w_r_shifted_eval", "w_o_shifted_eval", "q_c_eval", "q_mimc_coefficient_eval"]  # waffle_types.hpp:39-43
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


class MiMCComposer(StandardComposer):
    """waffle::MiMCComposer (composer/mimc_composer.hpp, mimc_composer.cpp:13-250): a MiMC round x_out = (x_in + k + c)^7 is ONE gate
    (w_l = k, w_r = (x_in + k + c)^3, w_o = x_in) whose result is the NEXT gate's output wire; gates that break the chain get a no-op
    gate in between to carry the pending output"""

    def __init__(self):
        super().__init__()
        self.q_mimc_coefficient, self.q_mimc_selector = [], []
        self.add_variable(0)
        self.zero_idx = 0
        self.current_output_wire = NO_WITNESS

    def _gate(self, a, b, c, q_m, q_l, q_r, q_o, q_c):  # every standard gate: mimc_composer.cpp:13-63
        if self.current_output_wire != NO_WITNESS:
            self.create_noop_gate()
        super()._gate(a, b, c, q_m, q_l, q_r, q_o, q_c)
        self.q_mimc_coefficient.append(0)
        self.q_mimc_selector.append(0)
        self.current_output_wire = NO_WITNESS

    def _zero_selectors(self, coefficient, selector):
        for q in (self.q_m, self.q_l, self.q_r, self.q_o, self.q_c):
            q.append(0)
        self.q_mimc_coefficient.append(coefficient % FR_MODULUS)
        self.q_mimc_selector.append(selector)

    # mimc_composer.cpp:65-90
    def create_mimc_gate(self, x_in_idx, x_cubed_idx, k_idx, x_out_idx, mimc_constant):
        if self.current_output_wire != NO_WITNESS and x_in_idx != self.current_output_wire:
            self.create_noop_gate()
        self.w_o.append(x_in_idx); self.w_l.append(k_idx); self.w_r.append(x_cubed_idx)
        self.current_output_wire = x_out_idx
        self._zero_selectors(mimc_constant, 1)
        self.wire_epicycles[k_idx].append((self.n, LEFT))
        self.wire_epicycles[x_cubed_idx].append((self.n, RIGHT))
        self.wire_epicycles[x_in_idx].append((self.n, OUTPUT))
        self.n += 1

    # mimc_composer.cpp:92-120
    def create_noop_gate(self):
        self._zero_selectors(0, 0)
        self.w_l.append(self.zero_idx); self.w_r.append(self.zero_idx)
        if self.current_output_wire != NO_WITNESS:
            self.w_o.append(self.current_output_wire)
            self.wire_epicycles[self.current_output_wire].append((self.n, OUTPUT))
            self.current_output_wire = NO_WITNESS
        else:
            self.w_o.append(self.zero_idx)
            self.wire_epicycles[self.zero_idx].append((self.n, OUTPUT))
        self.wire_epicycles[self.zero_idx].append((self.n, LEFT))
        self.wire_epicycles[self.zero_idx].append((self.n, RIGHT))
        self.n += 1

    # mimc_composer.cpp:170-250
    def preprocess(self):
        if self.current_output_wire != NO_WITNESS:  # close the chain: only the output wire of this gate is constrained
            self.w_o.append(self.current_output_wire); self.w_l.append(self.zero_idx); self.w_r.append(self.zero_idx)
            self._zero_selectors(0, 0)
            self.wire_epicycles[self.current_output_wire].append((self.n, OUTPUT))
            self.n += 1
            self.current_output_wire = NO_WITNESS
        n = self.n
        log2_n = n.bit_length() - 1
        if (1 << log2_n) != n:
            log2_n += 1
        new_n = 1 << log2_n
        pad = new_n - n
        cols = [c + [self.zero_idx] * pad for c in (self.w_l, self.w_r, self.w_o)]
        sel = [q + [0] * pad for q in (self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.q_mimc_selector, self.q_mimc_coefficient)]
        sigma = [np.arange(new_n, dtype=np.uint32) + np.uint32(t) for t in (LEFT, RIGHT, OUTPUT)]
        for cyc in self.wire_epicycles:
            for j, (gate, wire) in enumerate(cyc):
                nxt_gate, nxt_wire = cyc[0] if j == len(cyc) - 1 else cyc[j + 1]
                sigma[wire >> 30][gate] = np.uint32((nxt_gate + nxt_wire) & 0xFFFFFFFF)
        v = self.variables
        return {
            "n": new_n,
            "w_l": to_montgomery_limbs([v[i] for i in cols[0]]), "w_r": to_montgomery_limbs([v[i] for i in cols[1]]),
            "w_o": to_montgomery_limbs([v[i] for i in cols[2]]),
            "sigma_1_mapping": sigma[0], "sigma_2_mapping": sigma[1], "sigma_3_mapping": sigma[2],
            "q_m": to_montgomery_limbs(sel[0]), "q_l": to_montgomery_limbs(sel[1]), "q_r": to_montgomery_limbs(sel[2]),
            "q_o": to_montgomery_limbs(sel[3]), "q_c": to_montgomery_limbs(sel[4]),
            "q_mimc_selector": to_montgomery_limbs(sel[5]), "q_mimc_coefficient": to_montgomery_limbs(sel[6]),
        }


def mimc_circuit(num_gates, x0, k):
    """the MiMCComposer fixture circuit of oracle/plonk_driver.cpp (BB_CIRCUIT=mimc): num_gates - 2 rounds x <- (x + k + c_i)^7, then x + x0"""
    composer = MiMCComposer()
    k_idx = composer.add_variable(k)
    x, x_idx = x0 % FR_MODULUS, composer.add_variable(x0)
    x0_idx = x_idx
    for i in range(max(0, num_gates - 2)):
        c = (0x1000 + 7 * i) + ((i * i + 3) << 64) + (5 << 128)
        t0 = (x + k + c) % FR_MODULUS
        cubed = pow(t0, 3, FR_MODULUS)
        out = cubed * cubed * t0 % FR_MODULUS
        cubed_idx, out_idx = composer.add_variable(cubed), composer.add_variable(out)
        composer.create_mimc_gate(x_idx, cubed_idx, k_idx, out_idx, c)
        x, x_idx = out, out_idx
    s_idx = composer.add_variable(x + composer.get_variable(x0_idx))
    composer.create_add_gate(x_idx, x0_idx, s_idx, 1, 1, -1, 0)
    return composer


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


def zero_wire_circuit(num_gates, a0):
    """a * 0 = 0 in every gate: w_r and w_o are identically zero, so the wire commitments W_R and W_O are the point at infinity
    (mirror of build_zero_wire_circuit in oracle/plonk_driver.cpp; tests/golden/infinity_commitments.json)"""
    composer = StandardComposer()
    zero_idx = composer.add_variable(0)
    a = a0 % FR_MODULUS
    for _ in range(num_gates):
        ai = composer.add_variable(a)
        composer.create_mul_gate(ai, zero_idx, zero_idx, 1, -1, 0)
        a = (a * a + 1) % FR_MODULUS
    return composer


class _Circuit(C.Structure):
    _fields_ = [("n", C.c_size_t)] + [(k, C.c_void_p) for k in ("w_l", "w_r", "w_o", "sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping",
                                                               "q_m", "q_l", "q_r", "q_o", "q_c", "q_bl", "q_br", "q_bo",
                                                               "q_mimc_selector", "q_mimc_coefficient", "q_o_next")]


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
        """-> (120,) uint64: nine affine commitments, seven evaluations, five widget-dependent evaluations (waffle_types.hpp:18-45)"""
        out = np.zeros(120, dtype=np.uint64)
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_construct_proof(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def preprocess(self):
        """waffle::preprocess(prover): -> dict of the eight verification-key commitments, each (8,) uint64 affine"""
        out = np.zeros(96, dtype=np.uint64)
        self.gpu.lib.bbgpu_plonk_preprocess.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
        self.gpu._chk(self.gpu.lib.bbgpu_plonk_preprocess(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        names = VK_POINTS_EXTENDED if "q_o_next" in self._keep else VK_POINTS_BOOL if "q_bl" in self._keep else \
            VK_POINTS_MIMC if "q_mimc_selector" in self._keep else VK_POINTS
        return {k: out[8 * i:8 * i + 8] for i, k in enumerate(names)}

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


def proof_lines(n, proof, mimc=False, sequential=False):
    """the text form oracle/plonk_driver.cpp prints and tests/golden/plonk_proofs.json stores (with the MiMC widget: two more evaluations,
    with the sequential widget: w_o_shifted_eval)"""
    out = ["n %d" % n]
    for i, name in enumerate(PROOF_POINTS):
        out.append("%s.x %s" % (name, hex4(proof[8 * i:8 * i + 4])))
        out.append("%s.y %s" % (name, hex4(proof[8 * i + 4:8 * i + 8])))
    for i, name in enumerate(PROOF_EVALS):
        out.append("%s %s" % (name, hex4(proof[72 + 4 * i:76 + 4 * i])))
    if mimc or sequential:
        for name in ("w_o_shifted_eval", "q_mimc_coefficient_eval")[:2 if mimc else 1]:
            i = PROOF_EVALS_WIDGET.index(name)
            out.append("%s %s" % (name, hex4(proof[100 + 4 * i:104 + 4 * i])))
    return out


def proof_from_lines(lines):
    """inverse of proof_lines (standard part): -> (n, (120,) uint64)"""
    n = int(lines[0].split()[1])
    words = []
    for ln in lines[1:1 + 18 + 7]:
        h = ln.split()[1]
        words += [int(h[16 * (3 - k):16 * (4 - k)], 16) for k in range(4)]
    return n, np.array(words + [0] * 20, dtype=np.uint64)
