// The reference's own calling pattern for pippenger() -- scalar_multiplication.cpp:703-738: batched_scalar_multiplications cuts a job into one point
// range per OpenMP thread and every thread calls pippenger(scalars + off, points + 2 off, len) AT THE SAME TIME, then the partial sums are added
// (:755-761) -- driven through the shim's mangled symbols on the GPU: T OpenMP threads call pippenger() on sub-slices of one resident point table
// concurrently (plus fft() on buffers of their own), the sum of their results must be the point one call over the whole range returns.
// Built and run by tests/test_gpu_boundary.py::test_reference_calling_pattern_openmp_pippenger (g++ -fopenmp, links libbbshim.so / libbbgpu.so).
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../barretenberg_amd/shim/bb_abi.hpp"
#include "../../include/bbgpu.h"

using namespace barretenberg;

static uint64_t sm_state;
static uint64_t splitmix()
{
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 16;
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    omp_set_num_threads(T);
    sm_state = 42;
    uint64_t x[4] = { splitmix(), splitmix(), splitmix(), splitmix() & 0x0fffffffffffffffULL };
    g1::affine_element* table = static_cast<g1::affine_element*>(aligned_alloc(64, 2 * n * sizeof(g1::affine_element)));
    fr::field_t* scalars = static_cast<fr::field_t*>(aligned_alloc(32, n * sizeof(fr::field_t)));
    if (bbgpu_srs_generate(x, n, reinterpret_cast<uint64_t*>(table)) < 0) { // the synthetic SRS x^i G as the 2n-entry endomorphism table, resident from here on
        std::printf("FAIL srs_generate: %s\n", bbgpu_last_error());
        return 1;
    }
    for (size_t i = 0; i < n; i++) {
        for (int l = 0; l < 4; l++) scalars[i].data[l] = splitmix();
        scalars[i].data[3] &= 0x1fffffffffffffffULL; // any representative below 2^253
    }
    int fails = 0;
    for (int round = 0; round < 3; round++) {
        const g1::element whole = scalar_multiplication::pippenger(scalars, table, n, 0);
        std::vector<g1::element> part((size_t)T);
        std::vector<int> fft_ok((size_t)T, 1);
#pragma omp parallel for
        for (int t = 0; t < T; t++) {
            const size_t off = n * (size_t)t / (size_t)T, len = n * (size_t)(t + 1) / (size_t)T - off; // :718-726
            part[(size_t)t] = scalar_multiplication::pippenger(scalars + off, table + 2 * off, len, 0);
            // a transform of the thread's own in between: fft then ifft must return the input (canonical values in)
            const size_t m = (size_t)1 << (8 + t % 4);
            std::vector<fr::field_t> co(m), orig;
            for (size_t i = 0; i < m; i++) {
                for (int l = 0; l < 4; l++) co[i].data[l] = (uint64_t)(i * 1315423911u + (size_t)t * 2654435761u + (size_t)l * 97u + 1);
                co[i].data[3] &= 0x0fffffffffffffffULL;
            }
            orig = co;
            evaluation_domain dom;
            std::memset(static_cast<void*>(&dom), 0, sizeof(dom));
            dom.size = m;
            polynomial_arithmetic::fft(co.data(), dom);
            polynomial_arithmetic::ifft(co.data(), dom);
            // inputs below 2^252 < r are canonical: the round trip is the identity
            if (std::memcmp(co.data(), orig.data(), m * sizeof(fr::field_t))) fft_ok[(size_t)t] = 0;
            part[(size_t)t] = scalar_multiplication::pippenger(scalars + off, table + 2 * off, len, 0); // and once more after it
        }
        uint64_t sum[12];
        bbgpu_g1_sum(reinterpret_cast<const uint64_t*>(part.data()), (size_t)T, sum);
        if (std::memcmp(sum, &whole, 64)) { // x, y of the normalised points
            std::printf("FAIL round %d: the threads' partial sums do not add up to the one-call result\n", round);
            fails++;
        }
        for (int t = 0; t < T; t++)
            if (!fft_ok[(size_t)t]) {
                std::printf("FAIL round %d: fft / ifft round trip of thread %d\n", round, t);
                fails++;
            }
    }
    if (fails) std::printf("FAILED %d\n", fails);
    else std::printf("ALL OK n=%zu threads=%d\n", n, T);
    free(table);
    free(scalars);
    return fails ? 1 : 0;
}
