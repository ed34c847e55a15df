// selftest.hip -- known-answer entry points for the DEVICE field and group layer (fe.hpp with the gfx950 asm products,
// g1.hpp), so that the arithmetic every kernel is built from is pinned directly against the reference's vectors
// (tests/golden/field_ops.json, g1_ops.json, reference_kats.json <- test/test_fq.cpp:51-133, test_fr.cpp:51-88,
// test_g1.cpp:41-122) and not only through end-to-end MSM / NTT / proof parity.  One lane per case; inputs and outputs in the
// reference's memory format (4 x u64 Montgomery-2^256 limbs; Jacobian {x,y,z}, infinity = bit 63 of y limb 3).
// The lazy-bound cases drive the representation to its declared extremes (limbs up to 4 U, values up to 168 p < 2^261),
// which the value ranges NTT / MSM happen to produce do not reach.
#include <hip/hip_runtime.h>

#include <functional>

#include "bbgpu_internal.h"
#include "g1.hpp"
#include "g1_quad.hpp"

namespace bbgpu {
namespace {

template <class F> __device__ Fe<F, 1, 6> ld(const uint64_t* p)
{
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w[2 * i] = (uint32_t)p[i];
        w[2 * i + 1] = (uint32_t)(p[i] >> 32);
    }
    return unpack<F>(w);
}
template <class F, int L, int V> __device__ void st_canonical(uint64_t* p, const Fe<F, L, V>& a)
{
    uint32_t w[8];
    to_canonical(a, w);
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
// a product of two memory-form values carries 2^(256 + 256 - 261): one more product with 2^266 restores x y 2^256
template <class F, int L, int V> __device__ auto fix(const Fe<F, L, V>& t)
{
    return mul(t, fe_from<F>(F::M256_TO_M261));
}

// 28 a with tight limbs: value bound 6 * 28 = 168 = MAXV
template <class F> __device__ Fe<F, 1, 168> times28(const Fe<F, 1, 6>& A)
{
    const auto a2 = weak(add(A, A));
    const auto a4 = weak(add(a2, a2));
    const auto a8 = weak(add(a4, a4));
    const auto a16 = weak(add(a8, a8));
    return weak(add(weak(add(a16, a8)), a4));
}

// every op returns the reference-format (Montgomery-2^256, canonical) value named in the comment; a, b are the operands' residues
template <class F> __global__ void selftest_field_kernel(const uint64_t* a_in, const uint64_t* b_in, uint64_t* out, int n, int op)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const auto A = ld<F>(a_in + 4 * i), B = ld<F>(b_in + 4 * i);
    uint64_t* o = out + 4 * i;
    switch (op) {
    case BBGPU_SELFTEST_MUL: st_canonical(o, fix(mul(A, B))); break;                      // a b
    case BBGPU_SELFTEST_SQR: st_canonical(o, fix(sqr(A))); break;                         // a^2
    case BBGPU_SELFTEST_ADD: st_canonical(o, add(A, B)); break;                           // a + b
    case BBGPU_SELFTEST_SUB: st_canonical(o, sub(A, B)); break;                           // a - b
    case BBGPU_SELFTEST_NEG: st_canonical(o, neg(A)); break;                              // -a
    case BBGPU_SELFTEST_MUL_ADD: st_canonical(o, fix(mul_add(A, B, weak(add(A, B)), weak(sub(A, B))))); break; // a b + (a + b)(a - b)
    case BBGPU_SELFTEST_MUL_SUB: st_canonical(o, fix(mul_sub(A, B, weak(add(A, A)), B))); break;               // a b - 2 a b
    case BBGPU_SELFTEST_LAZY_LIMBS: {                                                     // 2a 3b: unnormalised limbs (2 U and 3 U) straight into the product,
        const auto X = add(A, A);                 // L = 2                                //        the largest column sums the multiplier admits (L1 L2 = 6)
        const auto Y = add(add(B, B), B);         // L = 3
        st_canonical(o, fix(mul(X, Y)));
        break;
    }
    case BBGPU_SELFTEST_LAZY_WEAK: {                                                      // 4a (b - a): limbs up to 4 U on both sides, renormalised by mul()
        const auto X = add(add(A, A), add(A, A)); // L = 4, V = 24
        const auto Y = sub(B, A);                 // L = 4, V = 13
        st_canonical(o, fix(mul(X, Y)));
        break;
    }
    case BBGPU_SELFTEST_LAZY_VALUE: {                                                     // 28 a: the value bound at its maximum (168 p < 2^261)
        st_canonical(o, mul(times28(A), fe_from<F>(F::ONE))); // times one (2^261): the same residue, through the multiplier
        break;
    }
    case BBGPU_SELFTEST_REDUCE: {                                                         // 28 a through reduce_value instead
        st_canonical(o, reduce_value(times28(A)));
        break;
    }
    case BBGPU_SELFTEST_SQR_LAZY: {                                                       // (2a - b)^2 with unnormalised limbs going into sqr
        const auto X = sub(add(A, A), B); // L = 5
        st_canonical(o, fix(sqr(X)));
        break;
    }
    case BBGPU_SELFTEST_ZERO_TESTS: {                                                     // limb 0: bit 0 is_zero_slow(a - b), bit 1 is_zero_mulout((a - b) a)
        const auto D = sub(A, B);
        const auto M = mul(D, A);
        o[0] = (is_zero_slow(D) ? 1u : 0u) | (is_zero_mulout(M) ? 2u : 0u);
        o[1] = o[2] = o[3] = 0;
        break;
    }
    case BBGPU_SELFTEST_MUL_ADDHI: {                                                      // a b - a: REDC(a b) + (K p - a), exact limbs out
        // operands as the accumulation has them: one factor in Montgomery-261 form (a table point), the other and the addend in memory form
        const auto r = mul_addhi_ip(A, m256_to_m261<F>(B), neg(A));
        st_canonical(o, r);
        break;
    }
    case BBGPU_SELFTEST_SQR_ADDHI: {                                                      // a^2 - (b + 2a), the addend with limbs up to 5 U
        const auto X = m256_to_m261<F>(A);                          // a 2^261
        const auto E = neg(add(m256_to_m261<F>(B), dbl(X)));        // K p - (b + 2a) 2^261
        st_canonical(o, m261_to_m256<F>(sqr_addhi(X, E)));
        break;
    }
    default: o[0] = o[1] = o[2] = o[3] = ~0ull;
    }
}

__device__ bool ld_jacobian(Xyzz& r, const uint64_t* j)
{
    if ((j[7] >> 63) & 1) {
        set_infinity(r);
        return true;
    }
    const auto X = m256_to_m261<Fq>(ld<Fq>(j)), Y = m256_to_m261<Fq>(ld<Fq>(j + 4)), Z = m256_to_m261<Fq>(ld<Fq>(j + 8));
    const auto ZZ = sqr(Z);
    r.x = X;
    r.y = Y;
    r.zz = ZZ;
    r.zzz = mul(ZZ, Z);
    return false;
}
// result as {X, Y, ZZ, ZZZ} in the reference's Montgomery form, canonical (x = X / ZZ, y = Y / ZZZ); ZZ = 0 <=> infinity
__device__ void st_xyzz(uint64_t* o, const Xyzz& p)
{
    uint32_t w[32];
    store_xyzz_m256(w, p);
    for (int i = 0; i < 16; i++) o[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

// p_in: n x 12 limbs (Jacobian), q_in: n x 12 limbs (Jacobian; the mixed addition reads only its affine x, y)
__global__ void selftest_g1_kernel(const uint64_t* p_in, const uint64_t* q_in, uint64_t* out, int n, int op)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz P, Q, R;
    ld_jacobian(P, p_in + 12 * i);
    switch (op) {
    case BBGPU_SELFTEST_G1_MADD:       // g1::mixed_add (group.hpp:219-322), the MSM's hot operation,
    case BBGPU_SELFTEST_G1_MADD_NEG: { // and with its conditional negation of the affine operand (group_impl_asm.tcc:71-153) taken
        AffineV<2> a;
        a.x = m256_to_m261<Fq>(ld<Fq>(q_in + 12 * i));
        a.y = m256_to_m261<Fq>(ld<Fq>(q_in + 12 * i + 4));
        R = P;
        madd(R, cond_neg_affine(a, op == BBGPU_SELFTEST_G1_MADD_NEG));
        break;
    }
    case BBGPU_SELFTEST_G1_ADD: // g1::add (:324-448)
        ld_jacobian(Q, q_in + 12 * i);
        add(R, P, Q);
        break;
    case BBGPU_SELFTEST_G1_DBL: // g1::dbl (:153-217)
        dbl(R, P);
        break;
    case BBGPU_SELFTEST_G1_DBL_AFFINE: { // the P + P branch of the mixed addition
        AffineV<2> a;
        a.x = m256_to_m261<Fq>(ld<Fq>(p_in + 12 * i));
        a.y = m256_to_m261<Fq>(ld<Fq>(p_in + 12 * i + 4));
        dbl_affine(R, a);
        break;
    }
    default: set_infinity(R);
    }
    st_xyzz(out + 16 * i, R);
}

// the quad addition of g1_quad.hpp: four lanes per case, lane l holding coordinate l of both operands
__global__ void selftest_g1_quad_kernel(const uint64_t* p_in, const uint64_t* q_in, uint64_t* out, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
    const uint32_t l = (uint32_t)t & 3u;
    if (i >= n) return; // whole quads leave together
    Xyzz P, Q;
    ld_jacobian(P, p_in + 12 * i);
    ld_jacobian(Q, q_in + 12 * i);
    const FqN pl = l == 0 ? P.x : (l == 1 ? P.y : (l == 2 ? P.zz : P.zzz));
    const FqN ql = l == 0 ? Q.x : (l == 1 ? Q.y : (l == 2 ? Q.zz : Q.zzz));
    const FqN r = quad_add(pl, ql, l);
    uint32_t w[8];
    to_canonical(m261_to_m256<Fq>(r), w);
    for (int k = 0; k < 4; k++) out[16 * i + 4 * l + k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
}

int run(const void* a, size_t a_bytes, const void* b, size_t b_bytes, void* out, size_t out_bytes, int n, const std::function<void(uint64_t*, uint64_t*, uint64_t*)>& launch)
{
    uint64_t *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = BBGPU_ERR_HIP;
    do {
        if (hipMalloc((void**)&da, a_bytes) != hipSuccess || hipMalloc((void**)&db, b_bytes) != hipSuccess || hipMalloc((void**)&dout, out_bytes) != hipSuccess) break;
        if (hipMemcpy(da, a, a_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(db, b, b_bytes, hipMemcpyHostToDevice) != hipSuccess) break;
        launch(da, db, dout);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
        if (hipMemcpy(out, dout, out_bytes, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = BBGPU_OK;
    } while (0);
    if (rc) set_error("selftest: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    (void)n;
    return rc;
}

} // namespace
} // namespace bbgpu

using namespace bbgpu;

#pragma GCC visibility push(default)
extern "C" {

int bbgpu_selftest_field(int field, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out)
{
    if (!a || !b || !out || n == 0 || n > (1u << 20) || (field != 0 && field != 1)) return BBGPU_ERR_ARG;
    if (bbgpu_device_count() == 0) {
        set_error("no HIP device available: libbbgpu has no CPU fallback");
        return BBGPU_ERR_HIP;
    }
    return run(a, n * 32, b, n * 32, out, n * 32, (int)n, [&](uint64_t* da, uint64_t* db, uint64_t* dout) {
        const int blocks = (int)((n + 63) / 64);
        if (field == 0) selftest_field_kernel<FqP><<<blocks, 64>>>(da, db, dout, (int)n, op);
        else selftest_field_kernel<FrP><<<blocks, 64>>>(da, db, dout, (int)n, op);
    });
}

int bbgpu_selftest_g1(int op, const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out)
{
    if (!p || !q || !out || n == 0 || n > (1u << 20)) return BBGPU_ERR_ARG;
    if (bbgpu_device_count() == 0) {
        set_error("no HIP device available: libbbgpu has no CPU fallback");
        return BBGPU_ERR_HIP;
    }
    return run(p, n * 96, q, n * 96, out, n * 128, (int)n, [&](uint64_t* dp, uint64_t* dq, uint64_t* dout) {
        if (op == BBGPU_SELFTEST_G1_QUAD_ADD) selftest_g1_quad_kernel<<<(int)((4 * n + 63) / 64), 64>>>(dp, dq, dout, (int)n);
        else selftest_g1_kernel<<<(int)((n + 63) / 64), 64>>>(dp, dq, dout, (int)n, op);
    });
}

} // extern "C"
#pragma GCC visibility pop
