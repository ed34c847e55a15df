#!/usr/bin/env python3
"""Generates barretenberg_amd/csrc/fe_mont_gfx950.h: the 9x29-bit Montgomery product (R = 2^261) of fe.hpp as ONE
hand-scheduled gfx950 instruction sequence per multiplication (mul, sqr, a*b + c*d).

Why: hipcc's schedule of the same C++ (fe.hpp mul_raw/sqr_raw/mul2_raw) starts every column's sum at zero and adds the
previous column's carry with a separate 64-bit addition (v_lshl_add_u64: 17 per multiplication, 144 in one mixed
addition); here the carry is the addend of the column's first v_mad_u64_u32 and one accumulator walks all 18 columns.
Per multiplication: 162 v_mad_u64_u32 + 9 v_mul_lo_u32 + 17 v_and_b32 + 17 v_lshrrev_b64 + 1 v_alignbit_b32 = 206
instructions (the compiler: 277).  Semantics are those of fe.hpp (field_impl_int128.tcc:72-137 in the reference, other radix).

The accumulator pair and one scratch register are FIXED physical VGPRs (clobbers): AMDGPU inline asm has no
sub-register operand modifier, and the low word of the 64-bit accumulator feeds v_mul_lo_u32 / v_and_b32.

Run: python tools/gen_mont_asm.py > barretenberg_amd/csrc/fe_mont_gfx950.h
"""
NL = 9
ACC = "v[2:3]"
ACC_LO = "v2"
ACC_HI = "v3"
TMP = "v4"
CLOBBERS = '"v2", "v3", "v4", "vcc"'
MASK = "0x1fffffff"


def mad(x, y, first):
    return f"v_mad_u64_u32 {ACC}, vcc, {x}, {y}, {'0' if first else ACC}"


# Register roles.  Default ("r"): the nine outputs double as the quotient digits m[0..8] -- m[j] is dead from column j + 9 on, which is
# when out[j] is written.  In-place forms (q = "m", o = the consumed operand): the result takes the registers of one OPERAND instead
# (operand limb j is last read in column j + 8, out[j] is written at the end of column j + 9) and the quotient digits live in nine
# scratch registers.  Same instructions, same count; what changes is where the result lands: a loop-carried value multiplied in place
# (ZZ3 = ZZ1 * PP in the bucket accumulation) stays in its registers and the loop's back edge needs no copies.
def low_column(ins, k, prods, q="r"):
    """prods: list of (x, y) operand names of the a*b-type products of column k"""
    first = (k == 0)
    for (x, y) in prods:
        ins.append(mad(x, y, first))
        first = False
    for i in range(k):
        ins.append(mad(f"%[{q}{i}]", f"%[p{k - i}]", False))
    ins.append(f"v_mul_lo_u32 {TMP}, {ACC_LO}, %[pinv]")
    ins.append(f"v_and_b32 %[{q}{k}], {MASK}, {TMP}")
    ins.append(mad(f"%[{q}{k}]", "%[p0]", False))
    ins.append(f"v_lshrrev_b64 {ACC}, 29, {ACC}")


def high_column(ins, k, prods, q="r", o="r", e=None):
    for (x, y) in prods:
        ins.append(mad(x, y, False))
    for i in range(k - (NL - 1), NL):
        ins.append(mad(f"%[{q}{i}]", f"%[p{k - i}]", False))
    # "addhi" forms (e = name of a third operand): out = REDC(a b) + e, i.e. e * 2^261 is added to the double-width product -- limb j of e goes
    # into column j + 9 as one more multiply-add (by the constant 1, an SGPR: the cheaper form of the instruction).  The sum then comes out of
    # the product's own carry chain with exact limbs: a difference U - X needs no separate subtraction and renormalisation (fe.hpp mul_addhi).
    if e is not None:
        ins.append(mad(f"%[{e}{k - NL}]", "%[one]", False))
    # r[k-9] is dead as a Montgomery quotient digit from column k-1 on (m[i] is last used in column i+8)
    if k < 2 * NL - 2:
        ins.append(f"v_and_b32 %[{o}{k - NL}], {MASK}, {ACC_LO}")
        ins.append(f"v_lshrrev_b64 {ACC}, 29, {ACC}")
    else:
        # last column: limb 7 and the (unmasked) top limb; r8 = m[8] (or the operand's limb 8) was an operand of this column's mads
        ins.append(f"v_and_b32 %[{o}7], {MASK}, {ACC_LO}")
        ins.append(f"v_alignbit_b32 %[{o}8], {ACC_HI}, {ACC_LO}, 29")
        if e is not None:
            ins.append(f"v_add_u32 %[{o}8], %[{o}8], %[{e}8]")


def gen_mul(q="r", o="r", e=None):
    ins = []
    for k in range(2 * NL - 1):
        prods = [(f"%[a{i}]", f"%[b{k - i}]") for i in range(NL) if 0 <= k - i < NL]
        if k < NL:
            low_column(ins, k, prods, q)
        else:
            high_column(ins, k, prods, q, o, e)
    return ins


def gen_mul2(q="r", o="r"):
    ins = []
    for k in range(2 * NL - 1):
        prods = []
        for i in range(NL):
            if 0 <= k - i < NL:
                prods.append((f"%[a{i}]", f"%[b{k - i}]"))
                prods.append((f"%[c{i}]", f"%[d{k - i}]"))
        if k < NL:
            low_column(ins, k, prods, q)
        else:
            high_column(ins, k, prods, q, o)
    return ins


def gen_sqr(e=None):
    ins = [f"v_lshlrev_b32 %[t{i}], 1, %[a{i}]" for i in range(NL - 1)]
    for k in range(2 * NL - 1):
        prods = []
        if k % 2 == 0:
            prods.append((f"%[a{k // 2}]", f"%[a{k // 2}]"))
        for i in range(NL):
            j = k - i
            if i < j < NL:
                prods.append((f"%[t{i}]", f"%[a{j}]"))
        if k < NL:
            low_column(ins, k, prods)
        else:
            high_column(ins, k, prods, e=e)
    return ins


def count(ins):
    c = {}
    for s in ins:
        c[s.split()[0]] = c.get(s.split()[0], 0) + 1
    return c


def emit_fn(name, sig, ins, outs, ins_v, temps, one=False):
    lines = []
    lines.append(f"// {name}: " + ", ".join(f"{v} {k}" for k, v in sorted(count(ins).items())) + f" = {len(ins)} instructions")
    lines.append(f"template <class F> __device__ __forceinline__ void {name}({sig})")
    lines.append("{")
    for t in temps:
        lines.append(f"    uint32_t {t};")
    lines.append("    asm(")
    for s in ins:
        lines.append(f'        "{s}\\n\\t"')
    lines.append("        : " + ", ".join(outs))
    lines.append("        : " + ", ".join(ins_v) + ",")
    lines.append("          " + ", ".join(f'[p{i}] "s"(F::P[{i}])' for i in range(NL)) + ', [pinv] "s"(F::PINV)' + (', [one] "s"(1u)' if one else ""))
    lines.append(f"        : {CLOBBERS});")
    lines.append("}")
    return "\n".join(lines)


def main():
    print("// GENERATED by tools/gen_mont_asm.py -- do not edit.")
    print("// Hand-scheduled gfx950 Montgomery products for fe.hpp (9 x 29-bit limbs, R = 2^261): one asm block per multiplication,")
    print("// the column carry is the addend of the column's first v_mad_u64_u32, one accumulator (" + ACC + ") walks the 18 columns.")
    print("// Device-only; the host build of fe.hpp keeps the C++ reference form of the same arithmetic (tests/cpp/test_fe_host.cpp")
    print("// and the device self-test bbgpu_selftest_field compare the two).")
    print("#pragma once")
    print("#include <stdint.h>")
    print("namespace bbgpu {")
    outs = [f'[r{i}] "=&v"(out[{i}])' for i in range(NL)]
    a_in = [f'[a{i}] "v"(a[{i}])' for i in range(NL)]
    b_in = [f'[b{i}] "v"(b[{i}])' for i in range(NL)]
    c_in = [f'[c{i}] "v"(c[{i}])' for i in range(NL)]
    d_in = [f'[d{i}] "v"(d[{i}])' for i in range(NL)]
    print(emit_fn("mul_raw_gfx950", "const uint32_t (&a)[9], const uint32_t (&b)[9], uint32_t (&out)[9]", gen_mul(), outs, a_in + b_in, []))
    print()
    t_out = [f'[t{i}] "=&v"(t{i})' for i in range(NL - 1)]
    print(emit_fn("sqr_raw_gfx950", "const uint32_t (&a)[9], uint32_t (&out)[9]", gen_sqr(), outs + t_out, a_in, [f"t{i}" for i in range(NL - 1)]))
    print()
    print(emit_fn("mul2_raw_gfx950", "const uint32_t (&a)[9], const uint32_t (&b)[9], const uint32_t (&c)[9], const uint32_t (&d)[9], uint32_t (&out)[9]",
                  gen_mul2(), outs, a_in + b_in + c_in + d_in, []))
    print()
    # in-place forms: the result replaces operand a (mul) / operand c (a*b + c*d); quotient digits in scratch registers
    m_tmp = [f"m{i}" for i in range(NL)]
    m_out = [f'[m{i}] "=&v"(m{i})' for i in range(NL)]
    a_io = [f'[a{i}] "+v"(a[{i}])' for i in range(NL)]
    c_io = [f'[c{i}] "+v"(c[{i}])' for i in range(NL)]
    print(emit_fn("mul_raw_inplace_gfx950", "uint32_t (&a)[9], const uint32_t (&b)[9]", gen_mul("m", "a"), a_io + m_out, b_in, m_tmp))
    print()
    print(emit_fn("mul2_raw_inplace_gfx950", "const uint32_t (&a)[9], const uint32_t (&b)[9], uint32_t (&c)[9], const uint32_t (&d)[9]",
                  gen_mul2("m", "c"), c_io + m_out, a_in + b_in + d_in, m_tmp))
    print()
    # "addhi" forms: REDC(a b) + e and REDC(a^2) + e with e's limbs added inside the product's carry chain (exact limbs out)
    e_in = [f'[e{i}] "v"(e[{i}])' for i in range(NL)]
    print(emit_fn("mul_addhi_raw_inplace_gfx950", "uint32_t (&a)[9], const uint32_t (&b)[9], const uint32_t (&e)[9]", gen_mul("m", "a", "e"), a_io + m_out, b_in + e_in, m_tmp, one=True))
    print()
    print(emit_fn("sqr_addhi_raw_gfx950", "const uint32_t (&a)[9], const uint32_t (&e)[9], uint32_t (&out)[9]", gen_sqr("e"), outs + t_out, a_in + e_in, [f"t{i}" for i in range(NL - 1)], one=True))
    print("} // namespace bbgpu")


if __name__ == "__main__":
    main()
