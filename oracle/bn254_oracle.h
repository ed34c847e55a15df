/*
 * bn254_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the arithmetic on barretenberg's MSM + NTT hot
 * path.  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check the HIP path; nothing in barretenberg_amd/ may
 * import, link or execute it.  Every function cites the reference file:line
 * (under /root/reference/src/barretenberg/) whose semantics it restates.
 *
 * Pinning: validated against (a) the known-answer vectors of the reference's
 * own tests (test/test_fq.cpp, test_fr.cpp, test_g1.cpp) and (b) outputs of
 * the reference itself compiled into oracle/_ref/ (see oracle/Makefile),
 * committed as fixtures under tests/golden/.
 *
 * Layouts match the reference ABI: field element = 4 x u64 little-endian limbs
 * (Montgomery form, R = 2^256); affine point = {x,y} = 8 x u64; Jacobian point
 * = {x,y,z} = 12 x u64; point at infinity <=> bit 63 of y limb 3.
 */
#ifndef BN254_ORACLE_H
#define BN254_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* field selector */
enum { ORC_FQ = 0, ORC_FR = 1 };

/* ---- prime field (fields/field_impl_int128.tcc, fields/field.hpp) ---- */
void orc_mul(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_sqr(int f, const uint64_t a[4], uint64_t r[4]);
void orc_mul_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_sqr_coarse(int f, const uint64_t a[4], uint64_t r[4]);
void orc_add(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_add_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_add_noreduce(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_sub(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_sub_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void orc_reduce_once(int f, const uint64_t a[4], uint64_t r[4]);
void orc_neg(int f, const uint64_t a[4], uint64_t r[4]);
void orc_to_mont(int f, const uint64_t a[4], uint64_t r[4]);
void orc_from_mont(int f, const uint64_t a[4], uint64_t r[4]);
void orc_pow(int f, const uint64_t a[4], const uint64_t e[4], uint64_t r[4]);
void orc_pow_small(int f, const uint64_t a[4], uint64_t e, uint64_t r[4]);
void orc_invert(int f, const uint64_t a[4], uint64_t r[4]);
void orc_batch_invert(int f, uint64_t* coeffs, size_t n);
void orc_mul_512(const uint64_t a[4], const uint64_t b[4], uint64_t r[8]);
void orc_get_root_of_unity(size_t degree_log2, uint64_t r[4]);
const uint64_t* orc_const(int f, const char* name); /* "modulus","one","r_squared","beta","generator","generator_inverse","alt_generator","root_of_unity" */

/* ---- endomorphism split + wNAF (fields/field.hpp:413-485, groups/wnaf.hpp:15-55) ---- */
void orc_split_endo(const uint64_t k[4], uint64_t k1[2], uint64_t k2[2]);
uint32_t orc_get_wnaf_bits(const uint64_t* scalar, size_t bits, size_t bit_position);
void orc_fixed_wnaf(const uint64_t scalar[2], uint32_t* wnaf, uint8_t* skew, size_t stride, size_t wnaf_bits);

/* ---- G1 (groups/group.hpp) ---- */
void orc_g1_set_infinity(uint64_t p[12]);
int orc_g1_is_infinity(const uint64_t* y_limbs_owner /* points at x; y at +4 */);
void orc_g1_dbl(const uint64_t p[12], uint64_t r[12]);
void orc_g1_mixed_add(const uint64_t p1[12], const uint64_t p2[8], uint64_t r[12]);
void orc_g1_add(const uint64_t p1[12], const uint64_t p2[12], uint64_t r[12]);
void orc_g1_normalize(const uint64_t p[12], uint64_t r[12]);
void orc_g1_batch_normalize(uint64_t* points, size_t n);
void orc_g1_neg_affine(const uint64_t p[8], uint64_t r[8]);
int orc_g1_on_curve_affine(const uint64_t p[8]);
void orc_g1_one_affine(uint64_t r[8]);
/* scalar multiplication of an affine point by a Montgomery-form scalar; result normalised
 * (z = one) or flagged infinity.  Mathematically equal to group.hpp:653-760. */
void orc_g1_scalar_mul(const uint64_t p[8], const uint64_t scalar_mont[4], uint64_t r[12]);

/* ---- MSM (curves/bn254/scalar_multiplication.cpp) ---- */
size_t orc_get_optimal_bucket_width(size_t num_points);
void orc_generate_point_table(const uint64_t* points, uint64_t* table, size_t n);
void orc_pippenger(const uint64_t* scalars_mont, const uint64_t* table, size_t n, size_t forced_bucket_width,
                   uint64_t out[12]);
/* batched entry: same job splitting as scalar_multiplication.cpp:650-772 for `threads` slices per job,
 * outputs normalised. */
struct orc_msm_job {
    const uint64_t* points;  /* 2n-entry endo table */
    const uint64_t* scalars; /* n Montgomery scalars */
    size_t num_elements;
    uint64_t output[12];
};
int orc_batched_msm(struct orc_msm_job* jobs, size_t num_jobs, size_t threads);
/* synthetic SRS: out[i] = x^i * G (affine, Montgomery), i < n; x given in Montgomery form. */
void orc_make_srs(const uint64_t x_mont[4], size_t n, uint64_t* out);

/* ---- NTT (polynomials/polynomial_arithmetic.cpp, evaluation_domain.cpp) ---- */
enum {
    ORC_FFT = 0,
    ORC_IFFT = 1,
    ORC_COSET_FFT = 2,
    ORC_COSET_IFFT = 3,
    ORC_FFT_WITH_CONSTANT = 4,
    ORC_IFFT_WITH_CONSTANT = 5,
    ORC_COSET_FFT_WITH_CONSTANT = 6
};
/* in place on coeffs[0..n); constant may be NULL for the kinds that take none. */
int orc_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant);
/* direct O(n) evaluation sum coeffs[i] z^i (polynomial_arithmetic.cpp:337-373), canonical result */
void orc_evaluate(const uint64_t* coeffs, const uint64_t z[4], size_t n, uint64_t r[4]);

/* ---- deterministic inputs: splitmix64 ---- */
uint64_t orc_splitmix64(uint64_t* state);
/* n scalars: 4 outputs per scalar, limb 3 &= 0x0fff..., then to Montgomery form */
void orc_random_scalars(uint64_t seed, size_t n, uint64_t* out_mont);

#ifdef __cplusplus
}
#endif
#endif
