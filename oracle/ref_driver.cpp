// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" wrapper (our code) around the *reference's own* compiled sources
// (curves/bn254/scalar_multiplication.cpp, polynomials/polynomial_arithmetic.cpp,
// polynomials/evaluation_domain.cpp under /root/reference/src/barretenberg), built by
// oracle/Makefile into oracle/_ref/libbbref.so (x86-64 asm path) and
// oracle/_ref/libbbref_portable.so (-DDISABLE_SHENANIGANS, the __int128 path).
// Used (a) to pin oracle/bn254_oracle.c, (b) to generate tests/golden/ fixtures
// (tools/gen_golden.py) and (c) as bench.py's cpu_baseline "reference" leg.
// No reference source text is copied here: this file only calls the reference's public API.
#include <barretenberg/curves/bn254/fq.hpp>
#include <barretenberg/curves/bn254/fr.hpp>
#include <barretenberg/curves/bn254/g1.hpp>
#include <barretenberg/curves/bn254/scalar_multiplication.hpp>
#include <barretenberg/groups/wnaf.hpp>
#include <barretenberg/polynomials/evaluation_domain.hpp>
#include <barretenberg/polynomials/polynomial_arithmetic.hpp>

#include <cstring>
#include <map>
#include <memory>
#include <omp.h>

using namespace barretenberg;

namespace {
inline fq::field_t ldq(const uint64_t* p) { fq::field_t r; memcpy(r.data, p, 32); return r; }
inline fr::field_t ldr(const uint64_t* p) { fr::field_t r; memcpy(r.data, p, 32); return r; }
inline void st(uint64_t* p, const uint64_t* d) { memcpy(p, d, 32); }
inline g1::element lde(const uint64_t* p) { g1::element e; memcpy(e.x.data, p, 32); memcpy(e.y.data, p + 4, 32); memcpy(e.z.data, p + 8, 32); return e; }
inline g1::affine_element lda(const uint64_t* p) { g1::affine_element e; memcpy(e.x.data, p, 32); memcpy(e.y.data, p + 4, 32); return e; }
inline void ste(uint64_t* p, const g1::element& e) { memcpy(p, e.x.data, 32); memcpy(p + 4, e.y.data, 32); memcpy(p + 8, e.z.data, 32); }

std::map<size_t, std::unique_ptr<evaluation_domain>>& domains()
{
    static std::map<size_t, std::unique_ptr<evaluation_domain>> d;
    return d;
}
const evaluation_domain& get_domain(size_t n)
{
    auto& d = domains();
    auto it = d.find(n);
    if (it == d.end()) {
        auto dom = std::make_unique<evaluation_domain>(n);
        dom->compute_lookup_table();
        it = d.emplace(n, std::move(dom)).first;
    }
    return *it->second;
}
} // namespace

#pragma GCC visibility push(default)
extern "C" {

int ref_uses_asm()
{
#ifdef DISABLE_SHENANIGANS
    return 0;
#else
    return 1;
#endif
}
int ref_max_threads() { return omp_get_max_threads(); }
void ref_set_threads(int n) { omp_set_num_threads(n); domains().clear(); }

// field ops; f = 0 fq, 1 fr.  op: 0 mul 1 sqr 2 add 3 sub 4 mul_coarse 5 add_coarse 6 sub_coarse 7 reduce_once
// 8 to_mont 9 from_mont 10 invert 11 neg 12 sqr_coarse
void ref_field_op(int f, int op, const uint64_t* a, const uint64_t* b, uint64_t* r)
{
    if (f == 0) {
        fq::field_t x = ldq(a), y = b ? ldq(b) : fq::zero, o = fq::zero;
        switch (op) {
        case 0: fq::__mul(x, y, o); break;
        case 1: fq::__sqr(x, o); break;
        case 2: fq::__add(x, y, o); break;
        case 3: fq::__sub(x, y, o); break;
        case 4: fq::__mul_with_coarse_reduction(x, y, o); break;
        case 5: fq::__add_with_coarse_reduction(x, y, o); break;
        case 6: fq::__sub_with_coarse_reduction(x, y, o); break;
        case 7: fq::reduce_once(x, o); break;
        case 8: fq::__to_montgomery_form(x, o); break;
        case 9: fq::__from_montgomery_form(x, o); break;
        case 10: fq::__invert(x, o); break;
        case 11: fq::__neg(x, o); break;
        case 12: fq::__sqr_with_coarse_reduction(x, o); break;
        }
        st(r, o.data);
    } else {
        fr::field_t x = ldr(a), y = b ? ldr(b) : fr::zero, o = fr::zero;
        switch (op) {
        case 0: fr::__mul(x, y, o); break;
        case 1: fr::__sqr(x, o); break;
        case 2: fr::__add(x, y, o); break;
        case 3: fr::__sub(x, y, o); break;
        case 4: fr::__mul_with_coarse_reduction(x, y, o); break;
        case 5: fr::__add_with_coarse_reduction(x, y, o); break;
        case 6: fr::__sub_with_coarse_reduction(x, y, o); break;
        case 7: fr::reduce_once(x, o); break;
        case 8: fr::__to_montgomery_form(x, o); break;
        case 9: fr::__from_montgomery_form(x, o); break;
        case 10: fr::__invert(x, o); break;
        case 11: fr::__neg(x, o); break;
        case 12: fr::__sqr_with_coarse_reduction(x, o); break;
        }
        st(r, o.data);
    }
}

void ref_split_endo(const uint64_t* k, uint64_t* k1, uint64_t* k2)
{
    fr::field_t kk = ldr(k), a = fr::zero, b = fr::zero;
    fr::split_into_endomorphism_scalars(kk, a, b);
    k1[0] = a.data[0]; k1[1] = a.data[1];
    k2[0] = b.data[0]; k2[1] = b.data[1];
}

void ref_fixed_wnaf(const uint64_t* scalar, uint32_t* wnaf, uint8_t* skew, size_t stride, size_t wnaf_bits)
{
    uint64_t s[2] = { scalar[0], scalar[1] };
    bool sk = false;
    wnaf::fixed_wnaf(s, wnaf, sk, stride, wnaf_bits);
    *skew = sk ? 1 : 0;
}

// group ops: 0 dbl(p1) 1 mixed_add(p1, affine p2) 2 add(p1, p2) 3 normalize(p1)
void ref_g1_op(int op, const uint64_t* p1, const uint64_t* p2, uint64_t* r)
{
    g1::element a = lde(p1), o = a;
    switch (op) {
    case 0: g1::dbl(a, o); break;
    case 1: { g1::affine_element b = lda(p2); g1::mixed_add(a, b, o); break; }
    case 2: { g1::element b = lde(p2); g1::add(a, b, o); break; }
    case 3: o = g1::normalize(a); break;
    }
    ste(r, o);
}

// affine scalar multiplication (group.hpp:653-760 path), scalar in Montgomery form; result normalised or infinity
void ref_g1_scalar_mul(const uint64_t* p, const uint64_t* scalar_mont, uint64_t* r)
{
    g1::affine_element a = lda(p);
    fr::field_t s = ldr(scalar_mont);
    g1::affine_element o = g1::group_exponentiation(a, s);
    memcpy(r, o.x.data, 32);
    memcpy(r + 4, o.y.data, 32);
    memcpy(r + 8, fq::one.data, 32);
}

size_t ref_get_optimal_bucket_width(size_t n) { return scalar_multiplication::get_optimal_bucket_width(n); }

// table must hold 2n affine points; points may alias table
void ref_generate_point_table(uint64_t* points, uint64_t* table, size_t n)
{
    scalar_multiplication::generate_pippenger_point_table((g1::affine_element*)points, (g1::affine_element*)table, n);
}

// buffers must be 32-byte aligned (numpy/posix_memalign on the caller side)
void ref_pippenger(uint64_t* scalars, uint64_t* table, size_t n, size_t forced_bucket_width, uint64_t* out)
{
    g1::element r = scalar_multiplication::pippenger((fr::field_t*)scalars, (g1::affine_element*)table, n, forced_bucket_width);
    ste(out, r);
}

// scalar_multiplication.cpp:142-262: PLAIN n-entry point table (n * 64 bytes), scalars are clobbered (test_scalar_multiplication.cpp:164-187)
void ref_pippenger_low_memory(uint64_t* scalars, uint64_t* points, size_t n, uint64_t* out)
{
    g1::element r = scalar_multiplication::pippenger_low_memory((fr::field_t*)scalars, (g1::affine_element*)points, n);
    ste(out, r);
}

// num jobs over the same n; outputs normalised (scalar_multiplication.cpp:650-772)
void ref_batched_msm(uint64_t** scalars, uint64_t** tables, size_t n, size_t num, uint64_t* outs)
{
    scalar_multiplication::multiplication_state* st_ =
        (scalar_multiplication::multiplication_state*)aligned_alloc(32, sizeof(scalar_multiplication::multiplication_state) * num);
    for (size_t i = 0; i < num; i++) {
        st_[i].points = (g1::affine_element*)tables[i];
        st_[i].scalars = (fr::field_t*)scalars[i];
        st_[i].num_elements = n;
    }
    scalar_multiplication::batched_scalar_multiplications(st_, num);
    for (size_t i = 0; i < num; i++) ste(outs + 12 * i, st_[i].output);
    free(st_);
}

// kind as in oracle/bn254_oracle.h (ORC_FFT ...); in place; coeffs 32-byte aligned
int ref_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant)
{
    const evaluation_domain& d = get_domain(n);
    fr::field_t c = constant ? ldr(constant) : fr::one;
    fr::field_t* p = (fr::field_t*)coeffs;
    switch (kind) {
    case 0: polynomial_arithmetic::fft(p, d); break;
    case 1: polynomial_arithmetic::ifft(p, d); break;
    case 2: polynomial_arithmetic::coset_fft(p, d); break;
    case 3: polynomial_arithmetic::coset_ifft(p, d); break;
    case 4: polynomial_arithmetic::fft_with_constant(p, d, c); break;
    case 5: polynomial_arithmetic::ifft_with_constant(p, d, c); break;
    case 6: polynomial_arithmetic::coset_fft_with_constant(p, d, c); break;
    default: return 2;
    }
    return 0;
}
// build (and cache) the domain + twiddle table for n ahead of a timed region
void ref_prepare_domain(size_t n) { (void)get_domain(n); }

void ref_evaluate(const uint64_t* coeffs, const uint64_t* z, size_t n, uint64_t* r)
{
    fr::field_t zz = ldr(z);
    fr::field_t o = polynomial_arithmetic::evaluate((const fr::field_t*)coeffs, zz, n);
    st(r, o.data);
}

// ---- the O(n) helpers between the transforms; buffers 32-byte aligned ---------------------------------
void ref_batch_invert(uint64_t* coeffs, size_t n) { fr::batch_invert((fr::field_t*)coeffs, n); }

// outputs are left coarse ([0, 2r)) by the reference on purpose (polynomial_arithmetic.cpp:580-588); f receives F(z)
void ref_kate_opening(const uint64_t* src, uint64_t* dest, const uint64_t* z, size_t n, uint64_t* f)
{
    fr::field_t zz = ldr(z);
    fr::field_t o = polynomial_arithmetic::compute_kate_opening_coefficients((const fr::field_t*)src, (fr::field_t*)dest, zz, n);
    st(f, o.data);
}
void ref_lagrange_l1_fft(uint64_t* l_1, size_t n_src, size_t n_target)
{
    polynomial_arithmetic::compute_lagrange_polynomial_fft((fr::field_t*)l_1, get_domain(n_src), get_domain(n_target));
}
void ref_divide_by_pseudo_vanishing(uint64_t* coeffs, size_t n_src, size_t n_target)
{
    polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial((fr::field_t*)coeffs, get_domain(n_src), get_domain(n_target));
}
void ref_pointwise_mul(const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    polynomial_arithmetic::mul((const fr::field_t*)a, (const fr::field_t*)b, (fr::field_t*)r, get_domain(n));
}
// {Z_H*(z), L_1(z), L_{n-1}(z)} (polynomial_arithmetic.cpp:594-626)
void ref_lagrange_evaluations(const uint64_t* z, size_t n, uint64_t* out12)
{
    fr::field_t zz = ldr(z);
    polynomial_arithmetic::lagrange_evaluations e = polynomial_arithmetic::get_lagrange_evaluations(zz, get_domain(n));
    st(out12, e.vanishing_poly.data);
    st(out12 + 4, e.l_1.data);
    st(out12 + 8, e.l_n_minus_1.data);
}
}
#pragma GCC visibility pop
