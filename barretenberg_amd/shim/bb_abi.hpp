// bb_abi.hpp -- the minimum of barretenberg's type system needed to DEFINE the reference's own symbols
// (Itanium mangling and x86-64 SysV layout) from outside its source tree.  Declarations only: names, template
// parameters, member order and alignments are an ABI necessity; no function bodies of the reference
// appear here.  Sources of the layouts: fields/field.hpp:19-22 (field_t, alignas(32)), groups/group.hpp:17-28
// (affine_element, element), polynomials/evaluation_domain.hpp:9-59 (member order),
// curves/bn254/scalar_multiplication.hpp:14-39 (the CPU algorithm's state structs), :88-94 (multiplication_state).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace barretenberg {

template <typename FieldParams> class field {
  public:
    struct field_t {
        alignas(32) uint64_t data[4];
    };
};
class FrParams;
class Bn254FqParams;
struct Bn254G1Params;
typedef field<Bn254FqParams> fq;
typedef field<FrParams> fr;

template <typename coordinate_field, typename subgroup_field, typename GroupParams> class group {
  public:
    struct affine_element {
        typename coordinate_field::field_t x;
        typename coordinate_field::field_t y;
    };
    struct element {
        typename coordinate_field::field_t x;
        typename coordinate_field::field_t y;
        typename coordinate_field::field_t z;
    };
};
typedef group<fq, fr, Bn254G1Params> g1;

class evaluation_domain {
  public:
    size_t size;
    size_t num_threads;
    size_t thread_size;
    size_t log2_size;
    size_t log2_thread_size;
    size_t log2_num_threads;
    fr::field_t root;
    fr::field_t root_inverse;
    fr::field_t domain;
    fr::field_t domain_inverse;
    fr::field_t generator;
    fr::field_t generator_inverse;

  private:
    std::vector<fr::field_t*> round_roots;
    std::vector<fr::field_t*> inverse_round_roots;
    fr::field_t* roots;
};

namespace scalar_multiplication {
struct multiplication_state {
    g1::affine_element* points;
    fr::field_t* scalars;
    size_t num_elements;
    g1::element output;
};
// the CPU algorithm's own state (scalar_multiplication.hpp:14-39): member order is the ABI
struct wnaf_runtime_state {
    uint64_t current_sign;
    uint64_t next_sign;
    size_t current_idx;
    size_t next_idx;
    size_t bits_per_wnaf;
    uint32_t* wnaf_iterator;
    uint32_t* wnaf_table;
    bool* skew_table;
};
struct multiplication_runtime_state {
    size_t num_points;
    size_t num_rounds;
    size_t num_buckets;
    size_t switch_point;
    g1::element* buckets;
    g1::affine_element addition_temporary;
    g1::element accumulator;
    g1::element running_sum;
};
void compute_next_bucket_index(wnaf_runtime_state& state);
void compute_wnaf_state(multiplication_runtime_state& state, wnaf_runtime_state& wnaf_state, fr::field_t* scalars, size_t num_initial_points,
                        fr::field_t* endo_scalars, size_t forced_bucket_width);
g1::element pippenger_internal(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, fr::field_t* endo_scalars,
                               size_t forced_bucket_width);
g1::element alt_pippenger_internal(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, fr::field_t* endo_scalars,
                                   size_t forced_bucket_width);
g1::element pippenger(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, size_t forced_bucket_width);
std::vector<g1::affine_element*> generate_pippenger_precompute_table(g1::affine_element* points, g1::affine_element* table, size_t num_points, size_t bits_per_bucket);
g1::element pippenger_internal_precomputed(fr::field_t* scalars, const std::vector<g1::affine_element*>& round_points, const size_t num_initial_points,
                                           fr::field_t* endo_scalars);
g1::element pippenger_precomputed(fr::field_t* scalars, const std::vector<g1::affine_element*>& round_points, const size_t num_initial_points);
void batched_scalar_multiplications(multiplication_state* mul_state, size_t num_batches);
void generate_pippenger_point_table(g1::affine_element* points, g1::affine_element* table, size_t num_points);
size_t get_optimal_bucket_width(const size_t num_points);
g1::element pippenger_low_memory(fr::field_t* scalars, g1::affine_element* points, size_t num_points);
g1::element alt_pippenger(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, size_t forced_bucket_width);
} // namespace scalar_multiplication

namespace polynomial_arithmetic {
void fft(fr::field_t* coeffs, const evaluation_domain& domain);
void ifft(fr::field_t* coeffs, const evaluation_domain& domain);
void coset_fft(fr::field_t* coeffs, const evaluation_domain& domain);
void coset_ifft(fr::field_t* coeffs, const evaluation_domain& domain);
void fft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& value);
void ifft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& value);
void coset_fft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& constant);
// co-resident functions of the same translation unit (polynomial_arithmetic.hpp:14-61) that the PLONK stack calls
struct lagrange_evaluations {
    fr::field_t vanishing_poly;
    fr::field_t l_1;
    fr::field_t l_n_minus_1;
};
fr::field_t evaluate(const fr::field_t* coeffs, const fr::field_t& z, const size_t n);
void copy_polynomial(fr::field_t* src, fr::field_t* dest, size_t num_src_coefficients, size_t num_target_coefficients);
void compute_lagrange_polynomial_fft(fr::field_t* l_1_coefficients, const evaluation_domain& src_domain, const evaluation_domain& target_domain);
void divide_by_pseudo_vanishing_polynomial(fr::field_t* coeffs, const evaluation_domain& src_domain, const evaluation_domain& target_domain);
fr::field_t compute_kate_opening_coefficients(const fr::field_t* src, fr::field_t* dest, const fr::field_t& z, const size_t n);
lagrange_evaluations get_lagrange_evaluations(const fr::field_t& z, const evaluation_domain& domain);
void compress_fft(const fr::field_t* src, fr::field_t* dest, const size_t current_size, const size_t compress_factor);
// the remaining externs of the translation unit (polynomial_arithmetic.cpp:37-127,129-264,317-335): no caller in the PLONK stack,
// but the reference's own benchmarks and tests link against them, and a replacement of the WHOLE translation unit defines them
void fft_inner_serial(fr::field_t* coeffs, const size_t domain_size, const std::vector<fr::field_t*>& root_table);
void fft_inner_parallel(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& root, const std::vector<fr::field_t*>& root_table);
void scale_by_generator(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& generator_start, const fr::field_t& generator_shift);
void compute_multiplicative_subgroup(const size_t log2_subgroup_size, const evaluation_domain& src_domain, fr::field_t* subgroup_roots);
void add(const fr::field_t* a_coeffs, const fr::field_t* b_coeffs, fr::field_t* r_coeffs, const evaluation_domain& domain);
void mul(const fr::field_t* a_coeffs, const fr::field_t* b_coeffs, fr::field_t* r_coeffs, const evaluation_domain& domain);
} // namespace polynomial_arithmetic

// layout probes (measured against the reference headers with sizeof/offsetof)
static_assert(sizeof(fr::field_t) == 32 && alignof(fr::field_t) == 32, "field_t");
static_assert(sizeof(g1::affine_element) == 64 && sizeof(g1::element) == 96, "g1 elements");
static_assert(sizeof(scalar_multiplication::multiplication_state) == 128, "multiplication_state");
static_assert(offsetof(scalar_multiplication::multiplication_state, output) == 32, "multiplication_state.output");
static_assert(sizeof(scalar_multiplication::wnaf_runtime_state) == 64 && sizeof(scalar_multiplication::multiplication_runtime_state) == 320, "CPU Pippenger state");
static_assert(offsetof(scalar_multiplication::multiplication_runtime_state, buckets) == 32 && offsetof(scalar_multiplication::multiplication_runtime_state, addition_temporary) == 64 &&
                  offsetof(scalar_multiplication::multiplication_runtime_state, accumulator) == 128 && offsetof(scalar_multiplication::multiplication_runtime_state, running_sum) == 224,
              "multiplication_runtime_state fields");
static_assert(sizeof(evaluation_domain) == 320, "evaluation_domain");
static_assert(offsetof(evaluation_domain, root) == 64 && offsetof(evaluation_domain, generator_inverse) == 224, "evaluation_domain fields");

} // namespace barretenberg
