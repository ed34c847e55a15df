// host_small.hpp -- the reference's "small sizes" contract: the Verifier's per-proof MSM over ~20 freshly built points
// (verifier.cpp:359-363) and the n <= 16 transforms of the smallest circuits (test_verifier.cpp:105-122) are answered on the host,
// without a device allocation or a kernel launch -- a GPU pays ~20 dependent launches for work a core finishes sooner.
// Plain restatements on host_g1.hpp / host_fr.hpp (product code, no oracle/): a signed 4-bit-window bucket method for the MSM
// (same group element as scalar_multiplication::pippenger, :457-476, whatever the algorithm: results are compared after
// normalisation) and the defining sums of polynomial_arithmetic::fft / ifft / coset_* (:266-315) for the transforms.
// Everything above BBGPU_HOST_MSM_MAX points / BBGPU_HOST_NTT_MAX elements runs on the GPU; bbgpu_set_host_thresholds(0, 0)
// turns the host path off (the GPU kernels are correct from n = 1, tests/test_gpu_parity.py runs both).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "host_fr.hpp"
#include "host_g1.hpp"

namespace bbgpu {
namespace host {

// acc += (x, y) affine, never infinity (group.hpp:311-312)   [madd-2008-s]
static inline Xyzz g1_madd(const Xyzz& p, const Fq& x, const Fq& y)
{
    if (g1_is_inf(p)) {
        Xyzz r;
        r.x = x;
        r.y = y;
        r.zz = FQ_ONE;
        r.zzz = FQ_ONE;
        return r;
    }
    Fq U2 = fq_mul(x, p.zz), S2 = fq_mul(y, p.zzz);
    Fq P = fq_sub(U2, p.x), R = fq_sub(S2, p.y);
    if (fq_is_zero(P)) {
        if (!fq_is_zero(R)) return g1_infinity();
        Xyzz a;
        a.x = x;
        a.y = y;
        a.zz = FQ_ONE;
        a.zzz = FQ_ONE;
        return g1_dbl(a);
    }
    Fq PP = fq_sqr(P), PPP = fq_mul(P, PP), Q = fq_mul(p.x, PP);
    Xyzz r;
    r.x = fq_sub(fq_sub(fq_sqr(R), PPP), fq_dbl(Q));
    r.y = fq_sub(fq_mul(R, fq_sub(Q, r.x)), fq_mul(p.y, PPP));
    r.zz = fq_mul(p.zz, PP);
    r.zzz = fq_mul(p.zzz, PPP);
    return r;
}

// sum_i k_i P_i for a handful of points.  scalars: n x 4 limbs, Montgomery, any representative below 2^256 (the prover hands [0, 2r));
// points: the 2n-entry endomorphism table (entry 2i = P_i is read, the endo partner is not needed).
static inline Xyzz msm_small(const uint64_t* scalars, const uint64_t* points_endo, size_t n, size_t stride = 16 /* u64 words between base points: 16 = the 2n-entry endo table, 8 = a plain n-entry table */)
{
    constexpr int C = 4, W = 64, NB = 1 << (C - 1); // 64 signed 4-bit windows cover 256 bits; digits in [-8, 8]
    std::vector<int8_t> digit(n * W);
    for (size_t i = 0; i < n; i++) {
        Fr k;
        memcpy(k.d, scalars + 4 * i, 32);
        k = fr_from_mont(k); // canonical integer < r < 2^254
        int carry = 0;
        for (int w = 0; w < W; w++) {
            int d = (int)((k.d[w >> 4] >> ((w & 15) * 4)) & 15) + carry;
            carry = d > NB;
            if (carry) d -= 2 * NB;
            digit[i * W + w] = (int8_t)d;
        }
    }
    std::vector<Fq> neg_y(n);
    const Fq zero = { { 0, 0, 0, 0 } };
    for (size_t i = 0; i < n; i++) {
        Fq y;
        memcpy(y.d, points_endo + stride * i + 4, 32);
        neg_y[i] = fq_sub(zero, y);
    }
    Xyzz acc = g1_infinity();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < C; k++) acc = g1_dbl(acc);
        Xyzz bucket[NB];
        for (auto& b : bucket) b = g1_infinity();
        bool any = false;
        for (size_t i = 0; i < n; i++) {
            const int d = digit[i * W + w];
            if (d == 0) continue;
            any = true;
            Fq x, y;
            memcpy(x.d, points_endo + stride * i, 32);
            memcpy(y.d, points_endo + stride * i + 4, 32);
            Xyzz& b = bucket[(d > 0 ? d : -d) - 1];
            b = g1_madd(b, x, d > 0 ? y : neg_y[i]);
        }
        if (!any) continue;
        Xyzz run = g1_infinity(), sum = g1_infinity();
        for (int j = NB - 1; j >= 0; --j) {
            run = g1_add(run, bucket[j]);
            sum = g1_add(sum, run);
        }
        acc = g1_add(acc, sum);
    }
    return acc;
}

// the fft family on a host buffer of n = 2^lg <= 64 elements by the defining sums (n^2 multiplications); kind = bbgpu_ntt_kind.
// in: any representative below 2^256; out: canonical.
static inline void ntt_small(uint64_t* coeffs, int lg, int kind, const uint64_t* constant)
{
    const size_t n = (size_t)1 << lg;
    const bool inverse = (kind == 1 || kind == 3 || kind == 5);
    const bool pre = (kind == 2 || kind == 6), post = (kind == 3), has_const = (kind == 4 || kind == 5 || kind == 6);
    Fr w = fr_root_of_unity(lg);
    if (inverse) w = fr_inv(w);
    const Fr g = fr_from_limbs(FrHostP::GEN5);
    std::vector<Fr> x(n), wp(n), out(n);
    Fr gp = fr_one();
    for (size_t j = 0; j < n; j++) {
        Fr v;
        memcpy(v.d, coeffs + 4 * j, 32);
        x[j] = fr_mul(v, pre ? gp : fr_one()); // also brings any 256-bit representative into [0, r)
        if (pre) gp = fr_mul(gp, g);
    }
    wp[0] = fr_one();
    for (size_t j = 1; j < n; j++) wp[j] = fr_mul(wp[j - 1], w);
    Fr scale = fr_one();
    if (inverse) scale = fr_inv(fr_from_u64((uint64_t)n));
    if (has_const) {
        Fr c;
        memcpy(c.d, constant, 32);
        scale = fr_mul(c, scale); // c may be any representative; scale < r
    }
    const Fr ginv = fr_inv(g);
    Fr gk = fr_one();
    for (size_t k = 0; k < n; k++) {
        Fr s = fr_zero();
        for (size_t j = 0; j < n; j++) s = fr_add(s, fr_mul(x[j], wp[(j * k) & (n - 1)]));
        s = fr_mul(s, scale);
        if (post) {
            s = fr_mul(s, gk);
            gk = fr_mul(gk, ginv);
        }
        out[k] = s;
    }
    for (size_t k = 0; k < n; k++) memcpy(coeffs + 4 * k, out[k].d, 32);
}

} // namespace host
} // namespace bbgpu
