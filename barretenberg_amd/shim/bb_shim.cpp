// bb_shim.cpp -- drop-in definitions of the reference's hot-path entry points over the C ABI of libbbgpu.so.
// Linking libbbshim.so (or this object) ahead of libbarretenberg.a's scalar_multiplication.o / polynomial_arithmetic.o
// makes an unmodified waffle::Prover run its MSMs and NTTs on the MI355X (INTEGRATION.md shows the link line and the
// objcopy recipe for the TU's other symbols).
//
//   scalar_multiplication::pippenger                     scalar_multiplication.cpp:457-476
//   scalar_multiplication::batched_scalar_multiplications scalar_multiplication.cpp:650-772
//   polynomial_arithmetic::{fft,ifft,coset_fft,coset_ifft,fft_with_constant,ifft_with_constant,coset_fft_with_constant}
//                                                         polynomial_arithmetic.cpp:266-315
// and the co-resident functions of the two translation units that the PLONK stack (prover, verifier, composer, reference
// string) calls, so that scalar_multiplication.o and polynomial_arithmetic.o can be dropped from the link entirely
// (oracle/Makefile target plonk_gpu_full does exactly that):
//   scalar_multiplication::generate_pippenger_point_table  scalar_multiplication.cpp:131-140
//   polynomial_arithmetic::evaluate :337-373, copy_polynomial :23-35, compute_lagrange_polynomial_fft :381-476,
//   divide_by_pseudo_vanishing_polynomial :478-560, compute_kate_opening_coefficients :562-591,
//   get_lagrange_evaluations :594-626, compress_fft :629-639
// Not defined: the CPU algorithm's internals (get_optimal_bucket_width, compute_wnaf_state, pippenger_internal, fft_inner_*,
// scale_by_generator, the experimental alt_pippenger family): nothing outside their own translation unit, tests and benches calls them.
//
// Error behaviour: the reference API has no error channel.  A failing GPU call prints the library's error
// and aborts: silently returning a wrong proof element is worse than stopping, and there is deliberately no CPU
// fallback in this path.
#include "bb_abi.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/bbgpu.h"

namespace {
[[noreturn]] void die(const char* what, int rc)
{
    std::fprintf(stderr, "bbgpu shim: %s failed (%d): %s\n", what, rc, bbgpu_last_error());
    std::abort();
}
} // namespace

namespace barretenberg {
namespace scalar_multiplication {

g1::element pippenger(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, size_t /*forced_bucket_width*/)
{
    // the bucket width is a tuning knob of the CPU algorithm (get_optimal_bucket_width); every width yields the same
    // group element, and the GPU path picks its own window size
    g1::element out;
    int rc = bbgpu_msm_g1(reinterpret_cast<const uint64_t*>(scalars), reinterpret_cast<const uint64_t*>(points), num_initial_points,
                          reinterpret_cast<uint64_t*>(&out));
    if (rc != BBGPU_OK) die("pippenger", rc);
    return out;
}

void batched_scalar_multiplications(multiplication_state* mul_state, size_t num_batches)
{
    static_assert(sizeof(multiplication_state) == sizeof(bbgpu_msm_job), "job layout");
    for (size_t i = 1; i < num_batches; ++i) {
        if (mul_state[i].num_elements != mul_state[0].num_elements) {
            std::printf("batched_scalar_multiplications err: each scalar mul must be same size.\n"); // :680-684
            return;
        }
    }
    int rc = bbgpu_msm_g1_batch(reinterpret_cast<bbgpu_msm_job*>(mul_state), num_batches);
    if (rc != BBGPU_OK) die("batched_scalar_multiplications", rc);
}

void generate_pippenger_point_table(g1::affine_element* points, g1::affine_element* table, size_t num_points)
{
    int rc = bbgpu_generate_point_table(reinterpret_cast<const uint64_t*>(points), reinterpret_cast<uint64_t*>(table), num_points);
    if (rc != BBGPU_OK) die("generate_pippenger_point_table", rc);
}

} // namespace scalar_multiplication

namespace polynomial_arithmetic {
namespace {
void run(fr::field_t* coeffs, const evaluation_domain& domain, int kind, const fr::field_t* c)
{
    int rc = bbgpu_ntt(reinterpret_cast<uint64_t*>(coeffs), domain.size, kind, c ? reinterpret_cast<const uint64_t*>(c->data) : nullptr);
    if (rc != BBGPU_OK) die("fft", rc);
}
} // namespace
void fft(fr::field_t* coeffs, const evaluation_domain& domain) { run(coeffs, domain, BBGPU_FFT, nullptr); }
void ifft(fr::field_t* coeffs, const evaluation_domain& domain) { run(coeffs, domain, BBGPU_IFFT, nullptr); }
void coset_fft(fr::field_t* coeffs, const evaluation_domain& domain) { run(coeffs, domain, BBGPU_COSET_FFT, nullptr); }
void coset_ifft(fr::field_t* coeffs, const evaluation_domain& domain) { run(coeffs, domain, BBGPU_COSET_IFFT, nullptr); }
void fft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& value) { run(coeffs, domain, BBGPU_FFT_WITH_CONSTANT, &value); }
void ifft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& value) { run(coeffs, domain, BBGPU_IFFT_WITH_CONSTANT, &value); }
void coset_fft_with_constant(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& constant) { run(coeffs, domain, BBGPU_COSET_FFT_WITH_CONSTANT, &constant); }

fr::field_t evaluate(const fr::field_t* coeffs, const fr::field_t& z, const size_t n)
{
    fr::field_t r;
    int rc = bbgpu_fr_evaluate(reinterpret_cast<const uint64_t*>(coeffs), n, z.data, r.data);
    if (rc != BBGPU_OK) die("evaluate", rc);
    return r;
}
void copy_polynomial(fr::field_t* src, fr::field_t* dest, size_t num_src_coefficients, size_t num_target_coefficients)
{
    // :23-35: copy, then zero the tail of the (longer) destination
    std::memcpy(dest, src, num_src_coefficients * sizeof(fr::field_t));
    if (num_target_coefficients > num_src_coefficients)
        std::memset(dest + num_src_coefficients, 0, (num_target_coefficients - num_src_coefficients) * sizeof(fr::field_t));
}
void compute_lagrange_polynomial_fft(fr::field_t* l_1_coefficients, const evaluation_domain& src_domain, const evaluation_domain& target_domain)
{
    int rc = bbgpu_lagrange_l1_fft(reinterpret_cast<uint64_t*>(l_1_coefficients), src_domain.size, target_domain.size);
    if (rc != BBGPU_OK) die("compute_lagrange_polynomial_fft", rc);
}
void divide_by_pseudo_vanishing_polynomial(fr::field_t* coeffs, const evaluation_domain& src_domain, const evaluation_domain& target_domain)
{
    int rc = bbgpu_divide_by_pseudo_vanishing(reinterpret_cast<uint64_t*>(coeffs), src_domain.size, target_domain.size);
    if (rc != BBGPU_OK) die("divide_by_pseudo_vanishing_polynomial", rc);
}
fr::field_t compute_kate_opening_coefficients(const fr::field_t* src, fr::field_t* dest, const fr::field_t& z, const size_t n)
{
    fr::field_t f;
    int rc = bbgpu_kate_opening(reinterpret_cast<const uint64_t*>(src), reinterpret_cast<uint64_t*>(dest), n, z.data, f.data);
    if (rc != BBGPU_OK) die("compute_kate_opening_coefficients", rc);
    return f;
}
lagrange_evaluations get_lagrange_evaluations(const fr::field_t& z, const evaluation_domain& domain)
{
    lagrange_evaluations r;
    static_assert(sizeof(lagrange_evaluations) == 96, "three field elements");
    int rc = bbgpu_lagrange_evaluations(z.data, domain.size, reinterpret_cast<uint64_t*>(&r));
    if (rc != BBGPU_OK) die("get_lagrange_evaluations", rc);
    return r;
}
void compress_fft(const fr::field_t* src, fr::field_t* dest, const size_t current_size, const size_t compress_factor)
{
    size_t log2_factor = 0;
    while (((size_t)1 << log2_factor) < compress_factor) ++log2_factor;
    const size_t new_size = current_size >> log2_factor;
    for (size_t i = 0; i < new_size; ++i) dest[i] = src[i << log2_factor]; // ascending: dest may overlap the front of src (:629-639)
}
} // namespace polynomial_arithmetic
} // namespace barretenberg
