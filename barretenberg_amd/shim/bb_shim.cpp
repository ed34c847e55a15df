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
// and the translation units' remaining externs, which only the reference's own tests and benchmarks call
// (bench_barretenberg.cpp:613-624 fft_inner_serial): get_optimal_bucket_width :21-81; fft_inner_serial :37-79, fft_inner_parallel
// :129-264 (forwarded to the GPU transform), scale_by_generator :81-102, compute_multiplicative_subgroup :104-127, add / mul :317-335
// -- O(n) host loops on csrc/host_fr.hpp, canonical outputs.
// plus pippenger_low_memory / alt_pippenger (same sum, same arguments).
// Also defined (SURVEY 8b "the MSM shim should define"): the CPU Pippenger's own machinery -- compute_wnaf_state (the GLV split and digit table of
// csrc/host_wnaf.hpp, host code), compute_next_bucket_index, pippenger_internal / alt_pippenger_internal (scalars already out of Montgomery form).
// and the pippenger_precomputed family (generate_pippenger_precompute_table fills the CPU layout of per-round tables as host code; the sums ignore them:
// the GPU keeps its own window tables resident).  Every extern of both replaced translation units is defined.
//
// Error behaviour (SURVEY 8b "C ABI underneath", "Errors"): the reference API has no error channel -- the signatures return void or a value,
// assert.hpp:13-23 compiles to nothing in release builds, batched_scalar_multiplications prints and returns (:680-684) -- so a GPU call
// that fails at RUN TIME (no device, an allocation refused on a shared GPU, a launch failure: BBGPU_ERR_HIP / _STATE / _SIZE) must not stop
// the prover.  The shim prints the library's error (the first time for every symbol, with a running count) and computes the same
// result with the library's own host code (bbgpu_host_*, csrc/host_fallback.hpp -- never oracle/), then carries on.  What still aborts: an
// argument error (BBGPU_ERR_ARG: a null pointer is a bug in the caller, not a condition of the machine), BBGPU_ERR_LOST (an in-place transform whose
// copy-back failed half way: the input is gone) and everything when
// BBGPU_SHIM_STRICT=1 is set (deployments that prefer stopping to running 100x slower; the GPU tests run that way so that a fallback can never
// stand in for a kernel).  bbshim_fallback_calls() returns how many calls were answered on the host.
#include "bb_abi.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/bbgpu.h"
#include "../csrc/host_fr.hpp"
#include "../csrc/host_g1.hpp"
#include "../csrc/host_wnaf.hpp"

#include <chrono>

#include <atomic>

namespace {
[[noreturn]] void die(const char* what, int rc)
{
    std::fprintf(stderr, "bbgpu shim: %s failed (%d): %s\n", what, rc, bbgpu_last_error());
    std::abort();
}
std::atomic<unsigned long long> g_fallbacks{ 0 };
// A GPU entry returned rc != BBGPU_OK.  Returns when the caller is to compute the result on the host; aborts for argument errors and in strict mode.
void gpu_failed(const char* what, int rc)
{
    static const bool strict = [] { const char* e = std::getenv("BBGPU_SHIM_STRICT"); return e && std::atoi(e) != 0; }();
    if (strict || rc == BBGPU_ERR_ARG || rc == BBGPU_ERR_LOST) die(what, rc); // LOST: an in-place buffer half overwritten -- no input left to compute from
    const unsigned long long k = g_fallbacks.fetch_add(1) + 1;
    static std::atomic<unsigned> logged{ 0 }; // a handful of lines, then one per thousand calls: the condition usually persists
    if (logged.fetch_add(1) < 8 || k % 1000 == 0)
        std::fprintf(stderr, "bbgpu shim: %s failed on the GPU (%d: %s) -- computing on the host (call %llu answered that way; BBGPU_SHIM_STRICT=1 aborts instead)\n",
                     what, rc, bbgpu_last_error(), k);
}
// the host computation itself cannot fail for arguments the GPU entry accepted; if it does, nothing is left to try
void host_must(const char* what, int rc)
{
    if (rc != BBGPU_OK) die(what, rc);
}

// Accounting of the drop-in path (BBGPU_SHIM_PROFILE=<file> or =1 for stderr; off by default, one branch per call when off): for every
// symbol the calls, the wall time spent inside it and the bytes its host buffers sent over the link each way, written as one JSON
// object when the process ends.  A caller's own timer around construct_proof() minus `total_ms` is then the reference's own host code;
// bytes / the link rate (tools/ubench/ubench_pcie: 56 GB/s either way on the MI355X boxes) is the part of `ms` that is PCIe.
struct ShimProfile {
    struct Row { const char* name; unsigned long long calls, up, down; double ms; };
    Row rows[48];
    int n = 0;
    const char* path = nullptr;
    ShimProfile() { path = std::getenv("BBGPU_SHIM_PROFILE"); if (path && !*path) path = nullptr; }
    Row& row(const char* name)
    {
        for (int i = 0; i < n; i++) if (rows[i].name == name) return rows[i]; // string literals: one address per call site
        if (n == 47) return rows[47];
        rows[n] = Row{ name, 0, 0, 0, 0.0 };
        return rows[n++];
    }
    bool written = false;
    void write(const char* tag, double caller_ms)
    {
        if (!path) return;
        FILE* f = std::strcmp(path, "1") ? std::fopen(path, written ? "a" : "w") : stderr;
        if (!f) return;
        written = true;
        double total = 0;
        unsigned long long up = 0, down = 0;
        for (int i = 0; i < n; i++) { total += rows[i].ms; up += rows[i].up; down += rows[i].down; }
        std::fprintf(f, "{\"region\": \"%s\", \"caller_ms\": %.3f, \"inside_shim_ms\": %.3f, \"h2d_bytes\": %llu, \"d2h_bytes\": %llu, \"symbols\": {", tag, caller_ms, total, up, down);
        for (int i = 0; i < n; i++)
            std::fprintf(f, "%s\"%s\": {\"calls\": %llu, \"ms\": %.3f, \"h2d_bytes\": %llu, \"d2h_bytes\": %llu}", i ? ", " : "", rows[i].name, rows[i].calls, rows[i].ms,
                         rows[i].up, rows[i].down);
        std::fprintf(f, "}}\n");
        if (f != stderr) std::fclose(f);
    }
    ~ShimProfile() { if (!written) write("process", 0.0); }
};
ShimProfile g_prof;
} // namespace
// for a caller that wants the accounting of ONE region (oracle/plonk_driver.cpp around construct_proof()): weak references on its side
extern "C" __attribute__((visibility("default"))) unsigned long long bbshim_fallback_calls(void) { return g_fallbacks.load(); }
extern "C" __attribute__((visibility("default"))) void bbshim_profile_reset(void) { g_prof.n = 0; }
extern "C" __attribute__((visibility("default"))) void bbshim_profile_write(const char* tag, double caller_ms) { g_prof.write(tag, caller_ms); }
namespace {
struct Prof {
    ShimProfile::Row* r = nullptr;
    std::chrono::steady_clock::time_point t0;
    Prof(const char* name, size_t up, size_t down)
    {
        if (!g_prof.path) return;
        r = &g_prof.row(name);
        r->calls++;
        r->up += up;
        r->down += down;
        t0 = std::chrono::steady_clock::now();
    }
    ~Prof()
    {
        if (!r) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        r->ms += ms;
        static const bool trace = std::getenv("BBGPU_SHIM_TRACE") != nullptr; // one line per call
        if (trace) std::fprintf(stderr, "bbshim %s %.3f ms\n", r->name, ms);
    }
};
} // namespace

namespace barretenberg {
namespace scalar_multiplication {

g1::element pippenger(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, size_t /*forced_bucket_width*/)
{
    // the bucket width is a tuning knob of the CPU algorithm (get_optimal_bucket_width); every width yields the same
    // group element, and the GPU path picks its own window size
    g1::element out;
    Prof prof("pippenger", num_initial_points * 32, 96); // + the point table the first time it is seen
    int rc = bbgpu_msm_g1(reinterpret_cast<const uint64_t*>(scalars), reinterpret_cast<const uint64_t*>(points), num_initial_points,
                          reinterpret_cast<uint64_t*>(&out));
    if (rc != BBGPU_OK) {
        gpu_failed("pippenger", rc);
        host_must("pippenger (host)", bbgpu_host_msm_g1(reinterpret_cast<const uint64_t*>(scalars), reinterpret_cast<const uint64_t*>(points), num_initial_points, 0,
                                                        reinterpret_cast<uint64_t*>(&out)));
    }
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
    Prof prof("batched_scalar_multiplications", num_batches ? num_batches * mul_state[0].num_elements * 32 : 0, num_batches * 96);
    int rc = bbgpu_msm_g1_batch(reinterpret_cast<bbgpu_msm_job*>(mul_state), num_batches);
    if (rc != BBGPU_OK) {
        gpu_failed("batched_scalar_multiplications", rc);
        for (size_t i = 0; i < num_batches; ++i) // every output again: the failing call may have written some of them
            host_must("batched_scalar_multiplications (host)",
                      bbgpu_host_msm_g1(reinterpret_cast<const uint64_t*>(mul_state[i].scalars), reinterpret_cast<const uint64_t*>(mul_state[i].points),
                                        mul_state[i].num_elements, 0, reinterpret_cast<uint64_t*>(&mul_state[i].output)));
    }
}

void generate_pippenger_point_table(g1::affine_element* points, g1::affine_element* table, size_t num_points)
{
    Prof prof("generate_pippenger_point_table", num_points * 64, num_points * 128);
    int rc = bbgpu_generate_point_table(reinterpret_cast<const uint64_t*>(points), reinterpret_cast<uint64_t*>(table), num_points);
    if (rc != BBGPU_OK) die("generate_pippenger_point_table", rc);
}

namespace {
// sum over a PLAIN n-entry point table (the reference's low-memory / precomputed entries apply beta themselves and never see the 2n-entry
// endomorphism table): reads exactly num_points * 64 bytes of `points`
g1::element msm_plain(const char* what, const fr::field_t* scalars, const g1::affine_element* points, size_t num_points)
{
    g1::element out;
    int rc = bbgpu_msm_g1_plain(reinterpret_cast<const uint64_t*>(scalars), reinterpret_cast<const uint64_t*>(points), num_points, reinterpret_cast<uint64_t*>(&out));
    if (rc != BBGPU_OK) {
        gpu_failed(what, rc);
        host_must(what, bbgpu_host_msm_g1(reinterpret_cast<const uint64_t*>(scalars), reinterpret_cast<const uint64_t*>(points), num_points, 1, reinterpret_cast<uint64_t*>(&out)));
    }
    return out;
}
fr::field_t* to_montgomery_copy(const char* what, const fr::field_t* plain, size_t n)
{
    using namespace bbgpu::host;
    fr::field_t* mont = static_cast<fr::field_t*>(aligned_alloc(32, sizeof(fr::field_t) * (n ? n : 1)));
    if (!mont) die(what, BBGPU_ERR_HIP);
    for (size_t i = 0; i < n; ++i) {
        Fr k;
        memcpy(k.d, plain[i].data, 32);
        k = fr_to_mont(k);
        memcpy(mont[i].data, k.d, 32);
    }
    return mont;
}
} // namespace

// pippenger_low_memory (:142-262): `points` is the PLAIN n-entry table (test_scalar_multiplication.cpp:164-187 allocates n * 64 bytes); the
// reference is allowed to clobber `scalars` (it leaves the split halves there), this one does not.
g1::element pippenger_low_memory(fr::field_t* scalars, g1::affine_element* points, size_t num_points) { return msm_plain("pippenger_low_memory", scalars, points, num_points); }
// alt_pippenger (bench_barretenberg.cpp:487-498): the 2n-entry endomorphism table, like pippenger
g1::element alt_pippenger(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, size_t forced_bucket_width)
{
    return pippenger(scalars, points, num_initial_points, forced_bucket_width);
}

// scalar_multiplication.cpp:21-81: the CPU algorithm's bucket width for n points (a cost-model table; the GPU path picks its own
// window size, callers outside the translation unit only read this for bookkeeping).  SURVEY a8: 2^20 -> 15, 2^16 -> 12, 8192 -> 10, 1 -> 1
size_t get_optimal_bucket_width(const size_t num_points)
{
    static const struct { size_t at_least, width; } table[] = {
        { 14617149, 21 }, { 2139094, 18 }, { 100000, 15 }, { 144834, 14 }, { 25067, 12 }, { 13926, 11 }, { 7659, 10 },
        { 2436, 9 },      { 376, 7 },      { 231, 6 },     { 97, 5 },      { 35, 4 },     { 10, 3 },     { 2, 2 },
    };
    for (const auto& row : table)
        if (num_points >= row.at_least) return row.width;
    return 1;
}

// ---- the CPU Pippenger's own machinery (no caller outside the translation unit in src/; the reference's tests and benches link it) --------------
// scalar_multiplication.cpp:83-88: the next digit-table entry, split into sign and bucket index
void compute_next_bucket_index(wnaf_runtime_state& state)
{
    const uint32_t entry = *state.wnaf_iterator;
    state.next_sign = (entry >> 31) & 1;
    state.next_idx = entry & 0x0fffffffU;
}
// :265-308: allocates the bucket array (2^c elements at infinity), the digit table (rounds x 2n entries, consecutive digits of one scalar 2n apart,
// most significant round first) and the skew bits, and fills them from the GLV halves of every scalar (written to endo_scalars[i].data[0..1] and
// [2..3]).  `scalars` are plain integers (the caller has left Montgomery form, :468-472).  The caller owns and frees the three arrays.
void compute_wnaf_state(multiplication_runtime_state& state, wnaf_runtime_state& wnaf_state, fr::field_t* scalars, size_t num_initial_points,
                        fr::field_t* endo_scalars, size_t forced_bucket_width)
{
    using namespace bbgpu::host;
    const size_t bits = forced_bucket_width > 0 ? forced_bucket_width : get_optimal_bucket_width(num_initial_points);
    state.num_points = 2 * num_initial_points;
    state.num_rounds = wnaf_size(bits + 1);
    state.num_buckets = (size_t)1 << bits;
    wnaf_state.bits_per_wnaf = bits + 1;
    auto round32 = [](size_t bytes) { return (bytes + 31) & ~(size_t)31; };
    state.buckets = static_cast<g1::element*>(aligned_alloc(32, round32(sizeof(g1::element) * state.num_buckets)));
    for (size_t i = 0; i < state.num_buckets; ++i) state.buckets[i].y.data[3] = 1ULL << 63; // set_infinity: the flag bit, nothing else (group.hpp:133-151)
    wnaf_state.wnaf_table = static_cast<uint32_t*>(aligned_alloc(32, round32(sizeof(uint32_t) * (state.num_rounds * state.num_points + 1))));
    wnaf_state.skew_table = static_cast<bool*>(aligned_alloc(32, round32(state.num_points + 1)));
    if (!state.buckets || !wnaf_state.wnaf_table || !wnaf_state.skew_table) die("compute_wnaf_state (allocation)", BBGPU_ERR_HIP);
    for (size_t i = 0; i < num_initial_points; ++i) {
        uint64_t k[4], k1[2], k2[2];
        memcpy(k, scalars[i].data, 32); // endo_scalars may alias scalars
        split_endo(k, k1, k2);
        endo_scalars[i].data[0] = k1[0]; endo_scalars[i].data[1] = k1[1];
        endo_scalars[i].data[2] = k2[0]; endo_scalars[i].data[3] = k2[1];
        fixed_wnaf(k1, &wnaf_state.wnaf_table[2 * i], wnaf_state.skew_table[2 * i], state.num_points, bits + 1);
        fixed_wnaf(k2, &wnaf_state.wnaf_table[2 * i + 1], wnaf_state.skew_table[2 * i + 1], state.num_points, bits + 1);
    }
    state.accumulator.y.data[3] = 1ULL << 63;
    wnaf_state.wnaf_iterator = wnaf_state.wnaf_table;
    compute_next_bucket_index(wnaf_state);
}
// :576-648 / alt: sum_i k_i P_i for scalars that have ALREADY left Montgomery form (`pippenger` converts and calls this, :468-474); `points` is the
// 2n-entry endomorphism table.  The sum is computed by the GPU path, which takes Montgomery scalars: one host multiplication by R^2 per scalar.
// endo_scalars (the CPU algorithm's scratch for the split scalars, usually the same array) is left as it is.
g1::element pippenger_internal(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, fr::field_t* /*endo_scalars*/,
                               size_t forced_bucket_width)
{
    if (num_initial_points == 0) return pippenger(scalars, points, 0, forced_bucket_width);
    fr::field_t* mont = to_montgomery_copy("pippenger_internal (allocation)", scalars, num_initial_points);
    g1::element out = pippenger(mont, points, num_initial_points, forced_bucket_width);
    free(mont);
    return out;
}
g1::element alt_pippenger_internal(fr::field_t* scalars, g1::affine_element* points, size_t num_initial_points, fr::field_t* endo_scalars,
                                   size_t forced_bucket_width)
{
    return pippenger_internal(scalars, points, num_initial_points, endo_scalars, forced_bucket_width);
}

// :90-129: per-round copies of a point table for the CPU algorithm's precomputed variant -- table[i * num_points + j] = 2^((c + 1)(i + 1)) * points[j],
// affine, i < rounds - 1; returns the tables most significant round first with `points` itself last.  Host code (host_g1.hpp doublings, one inversion per
// round): like the reference's, a one-off of rounds * (c + 1) doublings per point.  The GPU path keeps its own window tables and never reads these.
std::vector<g1::affine_element*> generate_pippenger_precompute_table(g1::affine_element* points, g1::affine_element* table, size_t num_points,
                                                                     size_t bits_per_bucket)
{
    using namespace bbgpu::host;
    const size_t rounds = wnaf_size(bits_per_bucket + 1);
    std::vector<Xyzz> cur(num_points);
    std::vector<Fq> den(num_points), pre(num_points);
    for (size_t j = 0; j < num_points; ++j) {
        memcpy(cur[j].x.d, points[j].x.data, 32);
        memcpy(cur[j].y.d, points[j].y.data, 32);
        cur[j].zz = FQ_ONE;
        cur[j].zzz = FQ_ONE;
    }
    for (size_t i = 0; i + 1 < rounds; ++i) {
        Fq run = FQ_ONE;
        for (size_t j = 0; j < num_points; ++j) {
            for (size_t k = 0; k < bits_per_bucket + 1; ++k) cur[j] = g1_dbl(cur[j]);
            pre[j] = run;
            den[j] = fq_mul(cur[j].zz, cur[j].zzz); // never zero: the points have prime order
            run = fq_mul(run, den[j]);
        }
        Fq inv = num_points ? fq_inv(run) : FQ_ONE;
        g1::affine_element* out = &table[i * num_points];
        for (size_t j = num_points; j-- > 0;) {
            const Fq dj = fq_mul(inv, pre[j]); // 1 / (zz zzz)
            inv = fq_mul(inv, den[j]);
            const Fq x = fq_mul(cur[j].x, fq_mul(dj, cur[j].zzz)), y = fq_mul(cur[j].y, fq_mul(dj, cur[j].zz));
            memcpy(out[j].x.data, x.d, 32);
            memcpy(out[j].y.data, y.d, 32);
            cur[j].x = x;
            cur[j].y = y;
            cur[j].zz = FQ_ONE;
            cur[j].zzz = FQ_ONE;
        }
    }
    std::vector<g1::affine_element*> result(rounds);
    result[rounds - 1] = points;
    for (size_t i = 0; i + 1 < rounds; ++i) result[rounds - 2 - i] = &table[i * num_points];
    return result;
}
// :478-574: sum_i k_i P_i where round_points[r][j], j < n, are PLAIN n-entry tables (the loop applies beta itself, :520-560) and
// round_points.back() is the caller's own `points` (test_scalar_multiplication.cpp:226-262 passes them without generate_pippenger_point_table).
// The internal form takes scalars that have already left Montgomery form (:485-488).
g1::element pippenger_internal_precomputed(fr::field_t* scalars, const std::vector<g1::affine_element*>& round_points, const size_t num_initial_points,
                                           fr::field_t* /*endo_scalars*/)
{
    fr::field_t* mont = to_montgomery_copy("pippenger_internal_precomputed (allocation)", scalars, num_initial_points);
    g1::element out = msm_plain("pippenger_internal_precomputed", mont, round_points.back(), num_initial_points);
    free(mont);
    return out;
}
g1::element pippenger_precomputed(fr::field_t* scalars, const std::vector<g1::affine_element*>& round_points, const size_t num_initial_points)
{
    return msm_plain("pippenger_precomputed", scalars, round_points.back(), num_initial_points);
}

} // namespace scalar_multiplication

namespace polynomial_arithmetic {
namespace {
void run(fr::field_t* coeffs, const evaluation_domain& domain, int kind, const fr::field_t* c)
{
    static const char* const names[] = { "fft", "ifft", "coset_fft", "coset_ifft", "fft_with_constant", "ifft_with_constant", "coset_fft_with_constant" };
    Prof prof(names[kind], domain.size * 32, domain.size * 32);
    int rc = bbgpu_ntt(reinterpret_cast<uint64_t*>(coeffs), domain.size, kind, c ? reinterpret_cast<const uint64_t*>(c->data) : nullptr);
    if (rc != BBGPU_OK) {
        gpu_failed(names[kind], rc);
        host_must(names[kind], bbgpu_host_ntt(reinterpret_cast<uint64_t*>(coeffs), domain.size, kind, c ? reinterpret_cast<const uint64_t*>(c->data) : nullptr));
    }
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
    Prof prof("evaluate", n * 32, 32);
    int rc = bbgpu_fr_evaluate(reinterpret_cast<const uint64_t*>(coeffs), n, z.data, r.data);
    if (rc != BBGPU_OK) {
        gpu_failed("evaluate", rc);
        host_must("evaluate (host)", bbgpu_host_fr_evaluate(reinterpret_cast<const uint64_t*>(coeffs), n, z.data, r.data));
    }
    return r;
}
void copy_polynomial(fr::field_t* src, fr::field_t* dest, size_t num_src_coefficients, size_t num_target_coefficients)
{
    // :23-35: copy, then zero the tail of the (longer) destination
    Prof prof("copy_polynomial", 0, 0);
    std::memcpy(dest, src, num_src_coefficients * sizeof(fr::field_t));
    if (num_target_coefficients > num_src_coefficients)
        std::memset(dest + num_src_coefficients, 0, (num_target_coefficients - num_src_coefficients) * sizeof(fr::field_t));
}
void compute_lagrange_polynomial_fft(fr::field_t* l_1_coefficients, const evaluation_domain& src_domain, const evaluation_domain& target_domain)
{
    Prof prof("compute_lagrange_polynomial_fft", 0, target_domain.size * 32);
    int rc = bbgpu_lagrange_l1_fft(reinterpret_cast<uint64_t*>(l_1_coefficients), src_domain.size, target_domain.size);
    if (rc != BBGPU_OK) {
        gpu_failed("compute_lagrange_polynomial_fft", rc);
        host_must("compute_lagrange_polynomial_fft (host)", bbgpu_host_lagrange_l1_fft(reinterpret_cast<uint64_t*>(l_1_coefficients), src_domain.size, target_domain.size));
    }
}
void divide_by_pseudo_vanishing_polynomial(fr::field_t* coeffs, const evaluation_domain& src_domain, const evaluation_domain& target_domain)
{
    Prof prof("divide_by_pseudo_vanishing_polynomial", target_domain.size * 32, target_domain.size * 32);
    int rc = bbgpu_divide_by_pseudo_vanishing(reinterpret_cast<uint64_t*>(coeffs), src_domain.size, target_domain.size);
    if (rc != BBGPU_OK) { // the failing call has not written coeffs: the device-to-host copy is its last step
        gpu_failed("divide_by_pseudo_vanishing_polynomial", rc);
        host_must("divide_by_pseudo_vanishing_polynomial (host)", bbgpu_host_divide_by_pseudo_vanishing(reinterpret_cast<uint64_t*>(coeffs), src_domain.size, target_domain.size));
    }
}
fr::field_t compute_kate_opening_coefficients(const fr::field_t* src, fr::field_t* dest, const fr::field_t& z, const size_t n)
{
    fr::field_t f;
    Prof prof("compute_kate_opening_coefficients", n * 32, n * 32);
    int rc = bbgpu_kate_opening(reinterpret_cast<const uint64_t*>(src), reinterpret_cast<uint64_t*>(dest), n, z.data, f.data);
    if (rc != BBGPU_OK) {
        gpu_failed("compute_kate_opening_coefficients", rc);
        host_must("compute_kate_opening_coefficients (host)", bbgpu_host_kate_opening(reinterpret_cast<const uint64_t*>(src), reinterpret_cast<uint64_t*>(dest), n, z.data, f.data));
    }
    return f;
}
lagrange_evaluations get_lagrange_evaluations(const fr::field_t& z, const evaluation_domain& domain)
{
    lagrange_evaluations r;
    static_assert(sizeof(lagrange_evaluations) == 96, "three field elements");
    Prof prof("get_lagrange_evaluations", 0, 0);
    int rc = bbgpu_lagrange_evaluations(z.data, domain.size, reinterpret_cast<uint64_t*>(&r));
    if (rc != BBGPU_OK) die("get_lagrange_evaluations", rc);
    return r;
}
void compress_fft(const fr::field_t* src, fr::field_t* dest, const size_t current_size, const size_t compress_factor)
{
    Prof prof("compress_fft", 0, 0);
    size_t log2_factor = 0;
    while (((size_t)1 << log2_factor) < compress_factor) ++log2_factor;
    const size_t new_size = current_size >> log2_factor;
    for (size_t i = 0; i < new_size; ++i) dest[i] = src[i << log2_factor]; // ascending: dest may overlap the front of src (:629-639)
}

// ---- the translation unit's remaining externs: host restatements (no caller on the prover path) ------------------------------------
namespace {
using bbgpu::host::Fr;
inline Fr ld(const fr::field_t& a)
{
    Fr r;
    std::memcpy(r.d, a.data, 32);
    return r;
}
inline void st(fr::field_t& a, const Fr& v) { std::memcpy(a.data, v.d, 32); }
inline Fr canon(const fr::field_t& a) { return bbgpu::host::fr_mul(ld(a), bbgpu::host::fr_one()); } // any 256-bit representative -> [0, r)
} // namespace

// :81-102  coeffs[i] *= generator_start * generator_shift^i
void scale_by_generator(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& generator_start, const fr::field_t& generator_shift)
{
    using namespace bbgpu::host;
    const Fr shift = canon(generator_shift);
    Fr work = canon(generator_start);
    for (size_t i = 0; i < domain.size; ++i) {
        st(coeffs[i], fr_mul(ld(coeffs[i]), work));
        work = fr_mul(work, shift);
    }
}
// :104-127  subgroup_roots[i] = g^n * w_k^i, k = 2^log2_subgroup_size, n = src_domain.size, g = the coset generator 5
void compute_multiplicative_subgroup(const size_t log2_subgroup_size, const evaluation_domain& src_domain, fr::field_t* subgroup_roots)
{
    using namespace bbgpu::host;
    const Fr root = fr_root_of_unity((int)log2_subgroup_size);
    Fr acc = fr_from_limbs(bbgpu::FrHostP::GEN5);
    for (size_t i = 0; i < src_domain.log2_size; ++i) acc = fr_sqr(acc);
    for (size_t i = 0; i < ((size_t)1 << log2_subgroup_size); ++i) {
        st(subgroup_roots[i], acc);
        acc = fr_mul(acc, root);
    }
}
// :317-335  pointwise over the domain
void add(const fr::field_t* a_coeffs, const fr::field_t* b_coeffs, fr::field_t* r_coeffs, const evaluation_domain& domain)
{
    for (size_t i = 0; i < domain.size; ++i) st(r_coeffs[i], bbgpu::host::fr_add(canon(a_coeffs[i]), canon(b_coeffs[i])));
}
void mul(const fr::field_t* a_coeffs, const fr::field_t* b_coeffs, fr::field_t* r_coeffs, const evaluation_domain& domain)
{
    for (size_t i = 0; i < domain.size; ++i) st(r_coeffs[i], bbgpu::host::fr_mul(ld(a_coeffs[i]), canon(b_coeffs[i])));
}
// :37-79  the single-threaded in-place transform over the caller's round-root table (bench-only in the reference, and kept a CPU
// symbol: SURVEY a20): bit-reversal, then log2 n rounds; root_table[s - 1][j] = w_{2m}^j for round m = 2^s.  Outputs canonical.
void fft_inner_serial(fr::field_t* coeffs, const size_t domain_size, const std::vector<fr::field_t*>& root_table)
{
    using namespace bbgpu::host;
    size_t log2_size = 0;
    while (((size_t)1 << log2_size) < domain_size) ++log2_size;
    for (size_t i = 0; i < domain_size; ++i) {
        size_t j = 0;
        for (size_t b = 0; b < log2_size; ++b) j |= ((i >> b) & 1) << (log2_size - 1 - b);
        if (i < j) {
            const fr::field_t t = coeffs[i];
            coeffs[i] = coeffs[j];
            coeffs[j] = t;
        }
    }
    for (size_t i = 0; i < domain_size; ++i) st(coeffs[i], canon(coeffs[i]));
    for (size_t m = 1, s = 0; m < domain_size; m *= 2, ++s) {
        for (size_t k = 0; k < domain_size; k += 2 * m) {
            for (size_t j = 0; j < m; ++j) {
                const Fr x = ld(coeffs[k + j]);
                const Fr t = m == 1 ? ld(coeffs[k + j + m]) : fr_mul(ld(root_table[s - 1][j]), ld(coeffs[k + j + m]));
                st(coeffs[k + j + m], fr_sub(x, t));
                st(coeffs[k + j], fr_add(x, t));
            }
        }
    }
}
// :129-264  the multi-threaded transform fft() and ifft() are built on: `root` says which of the two (the domain's root or its
// inverse; no scaling by 1/n in either case).  Forwarded to the GPU transform: ifft_with_constant(n) = the unscaled inverse transform.
void fft_inner_parallel(fr::field_t* coeffs, const evaluation_domain& domain, const fr::field_t& root, const std::vector<fr::field_t*>&)
{
    if (!std::memcmp(root.data, domain.root.data, 32)) {
        run(coeffs, domain, BBGPU_FFT, nullptr);
    } else if (!std::memcmp(root.data, domain.root_inverse.data, 32)) {
        fr::field_t n;
        st(n, bbgpu::host::fr_from_u64((uint64_t)domain.size));
        run(coeffs, domain, BBGPU_IFFT_WITH_CONSTANT, &n);
    } else {
        std::fprintf(stderr, "bbgpu shim: fft_inner_parallel with a root that is neither the domain's root nor its inverse\n");
        std::abort();
    }
}
} // namespace polynomial_arithmetic
} // namespace barretenberg
