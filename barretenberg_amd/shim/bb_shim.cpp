// bb_shim.cpp -- drop-in definitions of the reference's hot-path entry points over the C ABI of libbbgpu.so.
// Linking libbbshim.so (or this object) ahead of libbarretenberg.a's scalar_multiplication.o / polynomial_arithmetic.o
// makes an unmodified waffle::Prover run its MSMs and NTTs on the MI355X (INTEGRATION.md shows the link line and the
// objcopy recipe for the TU's other symbols).
//
//   scalar_multiplication::pippenger                     scalar_multiplication.cpp:457-476
//   scalar_multiplication::batched_scalar_multiplications scalar_multiplication.cpp:650-772
//   polynomial_arithmetic::{fft,ifft,coset_fft,coset_ifft,fft_with_constant,ifft_with_constant,coset_fft_with_constant}
//                                                         polynomial_arithmetic.cpp:266-315
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
} // namespace polynomial_arithmetic
} // namespace barretenberg
