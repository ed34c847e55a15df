// host_fallback.hpp -- the hot path's entry points computed on the host, for the ONE situation the drop-in boundary defines (SURVEY 8b
// "C ABI underneath": the reference's signatures return void / a value and have no error channel -- assert.hpp:13-23,
// scalar_multiplication.cpp:680-684 -- so the C++ shim must turn a failing GPU call into a CPU computation rather than stop the prover):
// no device, an allocation that failed on a shared GPU, a launch failure.  libbbgpu's GPU entry points never come here by themselves --
// they fail loudly (BBGPU_ERR_HIP) -- and nothing of this is on any measured path; shim/bb_shim.cpp calls the bbgpu_host_* entries that wrap
// these functions after it has logged the library's error, and BBGPU_SHIM_STRICT=1 makes it abort instead.
// Product code on host_fr.hpp / host_g1.hpp (no oracle/): plain textbook algorithms, written for correctness and a bearable speed
// (a few std::threads), not to compete with the reference's asm path:
//   msm_pippenger            sum_i k_i P_i        = scalar_multiplication::pippenger                      scalar_multiplication.cpp:457-476
//   ntt_radix2               the fft family       = polynomial_arithmetic::fft ... coset_fft_with_constant polynomial_arithmetic.cpp:266-315
//   poly_evaluate / kate_opening / lagrange_l1_fft / divide_by_pseudo_vanishing: the co-resident functions of the replaced translation unit
//                                                   that the PLONK stack calls                             polynomial_arithmetic.cpp:337-591
// Inputs may be any representative below 2^256, outputs are canonical, like the GPU entries.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/bbgpu.h"
#include "host_fr.hpp"
#include "host_g1.hpp"
#include "host_small.hpp"

namespace bbgpu {
namespace host {

// fn(lo, hi) over [0, count) on up to `threads` std::threads (the caller's included); small ranges stay on the caller's thread
template <class F> static inline void fallback_parallel(size_t count, size_t min_per_thread, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (const char* e = getenv("BBGPU_FALLBACK_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    size_t parts = std::min<size_t>(std::min<unsigned>(hw, 16u), count / std::max<size_t>(1, min_per_thread));
    if (parts <= 1) {
        fn((size_t)0, count);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(parts - 1);
    for (size_t p = 1; p < parts; p++) pool.emplace_back([=] { fn(count * p / parts, count * (p + 1) / parts); });
    fn((size_t)0, count / parts);
    for (auto& t : pool) t.join();
}

// ---- MSM: signed c-bit windows, one bucket set per window, windows dealt to threads, Horner over the window sums ---------------------------
// scalars: n x 4 limbs, Montgomery, any representative; points: base point i at points[stride_words * i] (16 = the 2n-entry endomorphism
// table of generate_pippenger_point_table, 8 = a plain n-entry table); never infinity (group.hpp:311-312).
static inline Xyzz msm_pippenger(const uint64_t* scalars, const uint64_t* points, size_t n, size_t stride_words)
{
    if (n == 0) return g1_infinity();
    if (n <= 32) return msm_small(scalars, points, n, stride_words);
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    const int c = std::min(15, std::max(4, lg - 3));
    const int W = 254 / c + 1; // canonical scalars are below 2^254; the carry out of the last full window lands in window W - 1
    const size_t NB = (size_t)1 << (c - 1);
    // digits, window-major: d in [-2^(c-1), 2^(c-1)]
    std::vector<int32_t> digit((size_t)W * n);
    fallback_parallel(n, 4096, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            Fr k;
            memcpy(k.d, scalars + 4 * i, 32);
            k = fr_from_mont(k); // the integer, canonical
            int carry = 0;
            for (int w = 0; w < W; w++) {
                const int bit = w * c, limb = bit >> 6, sh = bit & 63;
                uint64_t v = limb < 4 ? k.d[limb] >> sh : 0;
                if (sh + c > 64 && limb + 1 < 4) v |= k.d[limb + 1] << (64 - sh);
                int d = (int)(v & (((uint64_t)1 << c) - 1)) + carry;
                carry = d > (int)NB;
                if (carry) d -= (int)(2 * NB);
                digit[(size_t)w * n + i] = d;
            }
        }
    });
    std::vector<Xyzz> window_sum((size_t)W);
    fallback_parallel((size_t)W, 1, [&](size_t w0, size_t w1) {
        std::vector<Xyzz> bucket(NB);
        const Fq zero = { { 0, 0, 0, 0 } };
        for (size_t w = w0; w < w1; w++) {
            for (auto& b : bucket) b = g1_infinity();
            const int32_t* dg = &digit[w * n];
            for (size_t i = 0; i < n; i++) {
                const int d = dg[i];
                if (d == 0) continue;
                Fq x, y;
                memcpy(x.d, points + stride_words * i, 32);
                memcpy(y.d, points + stride_words * i + 4, 32);
                Xyzz& b = bucket[(size_t)(d > 0 ? d : -d) - 1];
                b = g1_madd(b, x, d > 0 ? y : fq_sub(zero, y));
            }
            Xyzz run = g1_infinity(), sum = g1_infinity(); // sum_j (j + 1) bucket[j] by the running sum from the top
            for (size_t j = NB; j-- > 0;) {
                run = g1_add(run, bucket[j]);
                sum = g1_add(sum, run);
            }
            window_sum[w] = sum;
        }
    });
    Xyzz acc = g1_infinity();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; k++) acc = g1_dbl(acc);
        acc = g1_add(acc, window_sum[(size_t)w]);
    }
    return acc;
}

// ---- the fft family, n = 2^lg: bit reversal, lg radix-2 decimation-in-time stages over a table of n / 2 powers of the root ------------------
// kind = bbgpu_ntt_kind; `constant` (Montgomery, any representative) is read for the *_with_constant kinds.
static inline void ntt_radix2(uint64_t* coeffs, int lg, int kind, const uint64_t* constant)
{
    const size_t n = (size_t)1 << lg;
    const bool inverse = (kind == BBGPU_IFFT || kind == BBGPU_COSET_IFFT || kind == BBGPU_IFFT_WITH_CONSTANT);
    const bool pre_coset = (kind == BBGPU_COSET_FFT || kind == BBGPU_COSET_FFT_WITH_CONSTANT), post_coset = (kind == BBGPU_COSET_IFFT);
    const bool has_const = (kind == BBGPU_FFT_WITH_CONSTANT || kind == BBGPU_IFFT_WITH_CONSTANT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    Fr* x = reinterpret_cast<Fr*>(coeffs);
    const Fr one = fr_one(), g = fr_from_limbs(FrHostP::GEN5);
    Fr w = fr_root_of_unity(lg);
    if (inverse) w = fr_inv(w);
    // powers of a base over a range, each thread starting from base^lo
    auto pow_at = [](Fr b, size_t e) {
        Fr acc = fr_one();
        for (; e; e >>= 1) {
            if (e & 1) acc = fr_mul(acc, b);
            b = fr_sqr(b);
        }
        return acc;
    };
    // load: canonicalise (a product with one brings any 256-bit representative into [0, r)), coset pre-scaling by g^i
    fallback_parallel(n, 8192, [&](size_t lo, size_t hi) {
        Fr gp = pre_coset ? pow_at(g, lo) : one;
        for (size_t i = lo; i < hi; i++) {
            x[i] = fr_mul(x[i], gp);
            if (pre_coset) gp = fr_mul(gp, g);
        }
    });
    for (size_t i = 0; i < n; i++) { // bit reversal
        size_t j = 0;
        for (int b = 0; b < lg; b++) j |= ((i >> b) & 1) << (lg - 1 - b);
        if (i < j) std::swap(x[i], x[j]);
    }
    std::vector<Fr> tw(std::max<size_t>(1, n / 2));
    fallback_parallel(n / 2, 8192, [&](size_t lo, size_t hi) {
        Fr p = pow_at(w, lo);
        for (size_t j = lo; j < hi; j++) {
            tw[j] = p;
            p = fr_mul(p, w);
        }
    });
    for (int s = 0; s < lg; s++) {
        const size_t m = (size_t)1 << s, step = n >> (s + 1); // butterflies (k + j, k + j + m), twiddle w_{2m}^j = tw[j * step]
        fallback_parallel(n / 2, 8192, [&](size_t lo, size_t hi) {
            for (size_t u = lo; u < hi; u++) {
                const size_t j = u & (m - 1), k = (u >> s) << (s + 1);
                const Fr a = x[k + j], t = j ? fr_mul(x[k + j + m], tw[j * step]) : x[k + j + m];
                x[k + j] = fr_add(a, t);
                x[k + j + m] = fr_sub(a, t);
            }
        });
    }
    Fr scale = one;
    if (inverse) scale = fr_inv(fr_from_u64((uint64_t)n));
    if (has_const) {
        Fr cst;
        memcpy(cst.d, constant, 32);
        scale = fr_mul(cst, scale);
    }
    if (inverse || has_const || post_coset) {
        const Fr ginv = fr_inv(g);
        fallback_parallel(n, 8192, [&](size_t lo, size_t hi) {
            Fr gk = post_coset ? fr_mul(scale, pow_at(ginv, lo)) : scale;
            for (size_t i = lo; i < hi; i++) {
                x[i] = fr_mul(x[i], gk);
                if (post_coset) gk = fr_mul(gk, ginv);
            }
        });
    }
}

// sum_i coeffs[i] z^i by Horner's rule from the top coefficient   (polynomial_arithmetic.cpp:337-373)
static inline Fr poly_evaluate(const uint64_t* coeffs, size_t n, const Fr& z_any)
{
    const Fr z = fr_mul(z_any, fr_one());
    const Fr* f = reinterpret_cast<const Fr*>(coeffs);
    Fr acc = fr_zero();
    for (size_t i = n; i-- > 0;) acc = fr_add(fr_mul(acc, z), fr_mul(f[i], fr_one()));
    return acc;
}

// dest = (F(X) - F(z)) / (X - z), returns F(z)   (polynomial_arithmetic.cpp:562-591).  Synthetic division from the top coefficient down:
// dest[n - 1] = 0, dest[i - 1] = F_i + z dest[i], F(z) = F_0 + z dest[0] -- the same quotient as the reference's bottom-up recurrence (which
// divides by -z at every step) since the division is exact.  dest may be src.
static inline Fr kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const Fr& z_any)
{
    if (n == 0) return fr_zero();
    const Fr z = fr_mul(z_any, fr_one()), one = fr_one();
    const Fr* f = reinterpret_cast<const Fr*>(src);
    Fr* q = reinterpret_cast<Fr*>(dest);
    Fr carry = fr_zero(); // dest[i] of the step before
    for (size_t i = n; i-- > 0;) {
        const Fr fi = fr_mul(f[i], one); // read before dest[i] is written: dest may alias src
        q[i] = carry;
        carry = fr_add(fi, fr_mul(z, carry));
    }
    return carry;
}

// L_1 on the coset g w_t^i of the target domain, n_t = 2^lt values   (polynomial_arithmetic.cpp:381-476):
// L_1(X) = (X^n - 1) / (n (X - 1)); on the coset X^n takes k = n_t / n values g^n w_k^(i mod k)
static inline void lagrange_l1_fft(uint64_t* out, int ls, int lt)
{
    const size_t nt = (size_t)1 << lt, k = (size_t)1 << (lt - ls);
    const Fr one = fr_one(), g = fr_from_limbs(FrHostP::GEN5), wt = fr_root_of_unity(lt), wk = fr_root_of_unity(lt - ls);
    Fr gn = g;
    for (int i = 0; i < ls; i++) gn = fr_sqr(gn);
    const Fr ninv = fr_inv(fr_from_u64((uint64_t)1 << ls));
    std::vector<Fr> numer(k);
    Fr p = gn;
    for (size_t j = 0; j < k; j++) {
        numer[j] = fr_mul(fr_sub(p, one), ninv);
        p = fr_mul(p, wk);
    }
    // denominators x_i - 1 inverted together (Montgomery's trick), x_i = g w_t^i; none is zero: g is not in the subgroup
    Fr* o = reinterpret_cast<Fr*>(out);
    std::vector<Fr> den(nt);
    Fr xi = g, run = one;
    for (size_t i = 0; i < nt; i++) {
        den[i] = fr_sub(xi, one);
        o[i] = run; // prefix product of the denominators before i
        run = fr_mul(run, den[i]);
        xi = fr_mul(xi, wt);
    }
    Fr inv = fr_inv(run);
    for (size_t i = nt; i-- > 0;) {
        const Fr di = fr_mul(inv, o[i]);
        inv = fr_mul(inv, den[i]);
        o[i] = fr_mul(di, numer[i & (k - 1)]);
    }
}

// coeffs[i] *= (x_i - w_n^-1) / (x_i^n - 1) on the coset x_i = g w_t^i of the target domain   (polynomial_arithmetic.cpp:478-560): division by the
// vanishing polynomial of the source domain without its last root
static inline void divide_by_pseudo_vanishing(uint64_t* coeffs, int ls, int lt)
{
    const size_t nt = (size_t)1 << lt, k = (size_t)1 << (lt - ls);
    const Fr one = fr_one(), g = fr_from_limbs(FrHostP::GEN5), wt = fr_root_of_unity(lt), wk = fr_root_of_unity(lt - ls);
    const Fr last = fr_inv(fr_root_of_unity(ls)); // w_n^(n - 1)
    Fr gn = g;
    for (int i = 0; i < ls; i++) gn = fr_sqr(gn);
    std::vector<Fr> inv(k);
    Fr p = gn;
    for (size_t j = 0; j < k; j++) {
        inv[j] = fr_inv(fr_sub(p, one));
        p = fr_mul(p, wk);
    }
    Fr* c = reinterpret_cast<Fr*>(coeffs);
    Fr xi = g;
    for (size_t i = 0; i < nt; i++) {
        c[i] = fr_mul(fr_mul(c[i], inv[i & (k - 1)]), fr_sub(xi, last));
        xi = fr_mul(xi, wt);
    }
}

} // namespace host
} // namespace bbgpu
