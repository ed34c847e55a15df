// Host-side unit test of barretenberg_amd/csrc/fe.hpp + g1.hpp (the exact code the HIP kernels run) against the
// oracle (oracle/bn254_oracle.c).  Test infrastructure: links liboracle.so.  Built and run by tests/test_host_field.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../barretenberg_amd/csrc/g1.hpp"
#include "../../oracle/bn254_oracle.h"

using namespace bbgpu;

static uint64_t st = 0x1234567;
static void rnd_canon(int f, uint64_t out[4])
{
    uint64_t raw[4];
    for (int i = 0; i < 4; i++) raw[i] = orc_splitmix64(&st);
    raw[3] &= 0x0fffffffffffffffULL;
    orc_to_mont(f, raw, out); // canonical residue
}
static void w8(const uint64_t a[4], uint32_t (&w)[8])
{
    for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)a[i]; w[2 * i + 1] = (uint32_t)(a[i] >> 32); }
}
static void u4(const uint32_t (&w)[8], uint64_t a[4])
{
    for (int i = 0; i < 4; i++) a[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static int fails = 0;
#define CHECK(c, msg) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, msg); fails++; } } while (0)
static bool eq4(const uint64_t a[4], const uint64_t b[4]) { return !memcmp(a, b, 32); }

template <class F> static void test_field(int f, const char* name)
{
    for (int it = 0; it < 2000; it++) {
        uint64_t a[4], b[4], c[4], want[4], got[4];
        rnd_canon(f, a); rnd_canon(f, b); rnd_canon(f, c);
        if (it == 0) memset(a, 0, 32);
        if (it == 1) { memcpy(a, orc_const(f, "modulus"), 32); a[0] -= 1; }
        uint32_t wa[8], wb[8], wc[8], wo[8];
        w8(a, wa); w8(b, wb); w8(c, wc);
        // inputs are canonical, so narrow the unpack() bound (V=6) to V=1 for the lazy chains below
        Fe<F, 1, 1> A, B, C;
        { auto a6 = unpack<F>(wa); auto b6 = unpack<F>(wb); auto c6 = unpack<F>(wc);
          for (int i = 0; i < 9; i++) { A.d[i] = a6.d[i]; B.d[i] = b6.d[i]; C.d[i] = c6.d[i]; } }
        // pack/unpack roundtrip
        { Fe<F, 1, 1> t; for (int i = 0; i < 9; i++) t.d[i] = A.d[i]; pack(t, wo); u4(wo, got); CHECK(eq4(got, a), "pack roundtrip"); }
        // mul: mont261(X,Y) * 2^266 / 2^261 == mont256(X,Y)
        { auto m = mul(mul(A, B), fe_from<F>(F::M256_TO_M261)); to_canonical(m, wo); u4(wo, got); orc_mul(f, a, b, want); CHECK(eq4(got, want), "mul"); }
        { auto m = mul(sqr(A), fe_from<F>(F::M256_TO_M261)); to_canonical(m, wo); u4(wo, got); orc_sqr(f, a, want); CHECK(eq4(got, want), "sqr"); }
        // add / sub / neg, lazily chained: (a + b - c) + (a - b) - (-c)
        { auto t = add(weak(sub(add(A, B), C)), weak(sub(A, B))); auto u = sub(weak(t), weak(neg(C)));
          to_canonical(u, wo); u4(wo, got);
          uint64_t x[4], y[4]; orc_add(f, a, b, x); orc_sub(f, x, c, x); orc_sub(f, a, b, y); orc_add(f, x, y, x); orc_neg(f, c, y); orc_sub(f, x, y, want);
          CHECK(eq4(got, want), "add/sub chain"); }
        // product of lazy operands: (a+b)*(a-c) ; (a-b)^2
        { auto m = mul(mul(add(A, B), sub(A, C)), fe_from<F>(F::M256_TO_M261)); to_canonical(m, wo); u4(wo, got);
          uint64_t x[4], y[4]; orc_add(f, a, b, x); orc_sub(f, a, c, y); orc_mul(f, x, y, want); CHECK(eq4(got, want), "lazy mul"); }
        { auto m = mul(sqr(sub(A, B)), fe_from<F>(F::M256_TO_M261)); to_canonical(m, wo); u4(wo, got);
          uint64_t x[4]; orc_sub(f, a, b, x); orc_sqr(f, x, want); CHECK(eq4(got, want), "lazy sqr"); }
        // big lazy value then reduce_value
        { auto t = add(weak(add(add(A, B), add(C, A))), weak(add(add(B, B), add(C, C)))); auto t2 = add(weak(t), weak(t)); auto t3 = add(weak(t2), weak(t2));
          auto r = reduce_value(t3); to_canonical(r, wo); u4(wo, got);
          uint64_t x[4], y[4]; orc_add(f, a, a, x); orc_add(f, b, b, y); orc_add(f, x, y, x); orc_add(f, x, b, x); orc_add(f, c, c, y); orc_add(f, x, y, x); orc_add(f, x, c, x);
          orc_add(f, x, x, x); orc_add(f, x, x, want); CHECK(eq4(got, want), "reduce_value"); to_canonical(t3, wo); u4(wo, got); CHECK(eq4(got, want), "to_canonical big"); }
        // conversions
        { auto m = m261_to_m256<F>(m256_to_m261<F>(unpack<F>(wa))); to_canonical(m, wo); u4(wo, got); CHECK(eq4(got, a), "m256<->m261"); }
        { auto m = mul(A, fe_from<F>(F::M256_TO_PLAIN)); to_canonical(m, wo); u4(wo, got); orc_from_mont(f, a, want); CHECK(eq4(got, want), "from montgomery"); }
        // zero tests
        { auto z = mul(sub(A, A), B); CHECK(is_zero_mulout(z), "zero mulout"); CHECK(is_zero_slow(sub(A, A)), "zero slow");
          if (it > 1) { CHECK(!is_zero_mulout(mul(sub(A, B), B)), "nonzero mulout"); } }
    }
    // non-canonical memory inputs (< 2^256): unpack then canonicalise == value mod p
    for (int it = 0; it < 200; it++) {
        uint64_t raw[4], want[4], got[4]; uint32_t w[8], wo[8];
        for (int i = 0; i < 4; i++) raw[i] = orc_splitmix64(&st);
        if (it < 4) memset(raw, 0xff, 32);
        w8(raw, w); to_canonical(unpack<F>(w), wo); u4(wo, got);
        // value mod p via oracle: from_mont(to_mont(raw))
        orc_to_mont(f, raw, want); orc_from_mont(f, want, want);
        CHECK(eq4(got, want), "unpack noncanonical");
    }
    printf("%s field ok (fails so far %d)\n", name, fails);
}

// XYZZ (montgomery-261) -> normalised affine in the reference's format, using the oracle's field ops
static void xyzz_to_ref_affine(const Xyzz& p, uint64_t out[12])
{
    uint32_t w[32]; store_xyzz_m256(w, p);
    uint64_t X[4], Y[4], ZZ[4], ZZZ[4];
    uint32_t t[8];
    memcpy(t, w, 32); u4(t, X); memcpy(t, w + 8, 32); u4(t, Y); memcpy(t, w + 16, 32); u4(t, ZZ); memcpy(t, w + 24, 32); u4(t, ZZZ);
    memset(out, 0, 96);
    if ((ZZ[0] | ZZ[1] | ZZ[2] | ZZ[3]) == 0) { out[7] = 1ULL << 63; return; }
    uint64_t i1[4], i2[4];
    orc_invert(ORC_FQ, ZZ, i1); orc_invert(ORC_FQ, ZZZ, i2);
    orc_mul(ORC_FQ, X, i1, out); orc_mul(ORC_FQ, Y, i2, out + 4); memcpy(out + 8, orc_const(ORC_FQ, "one"), 32);
}
static void ref_norm(const uint64_t p[12], uint64_t out[12])
{
    if (p[7] >> 63) { memset(out, 0, 96); out[7] = 1ULL << 63; return; }
    orc_g1_normalize(p, out);
}
static void load_ref_affine(AffineV<2>& a, const uint64_t p[8])
{
    uint32_t w[16];
    for (int i = 0; i < 8; i++) { w[2 * i] = (uint32_t)p[i]; w[2 * i + 1] = (uint32_t)(p[i] >> 32); }
    load_affine_m256(a, w);
}

static void test_group()
{
    const int N = 64;
    std::vector<uint64_t> srs(8 * N);
    uint64_t x[4]; rnd_canon(ORC_FR, x);
    orc_make_srs(x, N, srs.data());
    // device-format store/load roundtrip
    { AffineV<2> a; load_ref_affine(a, &srs[8]); uint32_t w[16]; store_affine_m261(w, a.x, a.y); AffineV<1> b; load_affine_m261(b, w);
      Xyzz p, q; from_affine(p, a); from_affine(q, b); uint64_t o1[12], o2[12]; xyzz_to_ref_affine(p, o1); xyzz_to_ref_affine(q, o2);
      CHECK(!memcmp(o1, o2, 96), "m261 store/load"); CHECK(!memcmp(o1, &srs[8], 64), "affine roundtrip"); }
    Xyzz acc; set_infinity(acc);
    uint64_t racc[12]; memset(racc, 0, 96); racc[7] = 1ULL << 63;
    uint64_t got[12], want[12];
    for (int i = 0; i < N; i++) {
        AffineV<2> a; load_ref_affine(a, &srs[8 * i]);
        bool negf = (i % 3) == 1;
        auto an = cond_neg_affine(a, negf);
        madd(acc, an);
        uint64_t pt[8]; memcpy(pt, &srs[8 * i], 64); if (negf) orc_neg(ORC_FQ, pt + 4, pt + 4);
        orc_g1_mixed_add(racc, pt, racc);
        xyzz_to_ref_affine(acc, got); ref_norm(racc, want); CHECK(!memcmp(got, want, 96), "madd chain");
        if (i % 5 == 0) { Xyzz d; dbl(d, acc); acc = d; orc_g1_dbl(racc, racc); xyzz_to_ref_affine(acc, got); ref_norm(racc, want); CHECK(!memcmp(got, want, 96), "dbl"); }
        if (i % 7 == 3) { Xyzz o; from_affine(o, a); Xyzz d; dbl(d, o); Xyzz s; add(s, acc, d); acc = s;
            uint64_t j[12]; memcpy(j, &srs[8 * i], 64); memcpy(j + 8, orc_const(ORC_FQ, "one"), 32); orc_g1_dbl(j, j); orc_g1_add(racc, j, racc);
            xyzz_to_ref_affine(acc, got); ref_norm(racc, want); CHECK(!memcmp(got, want, 96), "add"); }
    }
    // exceptional cases
    { AffineV<2> a; load_ref_affine(a, &srs[8 * 3]);
      Xyzz p; from_affine(p, a); madd(p, a);  // P + P
      uint64_t j[12]; memcpy(j, &srs[8 * 3], 64); memcpy(j + 8, orc_const(ORC_FQ, "one"), 32); orc_g1_dbl(j, j);
      xyzz_to_ref_affine(p, got); ref_norm(j, want); CHECK(!memcmp(got, want, 96), "madd P+P");
      Xyzz q; from_affine(q, a); auto an = cond_neg_affine(a, true); madd(q, an); CHECK(is_infinity(q), "madd P+(-P)");
      // non-trivial Z: acc (random) + its own affine form
      uint64_t aff[12]; xyzz_to_ref_affine(acc, aff); AffineV<2> b; load_ref_affine(b, aff);
      Xyzz r = acc; madd(r, b); Xyzz d; dbl(d, acc); uint64_t g2[12]; xyzz_to_ref_affine(r, got); xyzz_to_ref_affine(d, g2); CHECK(!memcmp(got, g2, 96), "madd acc+acc(aff)");
      Xyzz r2 = acc; madd(r2, cond_neg_affine(b, true)); CHECK(is_infinity(r2), "madd acc-acc");
      Xyzz s; add(s, acc, acc); xyzz_to_ref_affine(s, got); CHECK(!memcmp(got, g2, 96), "add P+P");
      Xyzz nb; from_affine(nb, cond_neg_affine(b, true)); Xyzz s2; add(s2, acc, nb); CHECK(is_infinity(s2), "add P+(-P)");
      Xyzz inf; set_infinity(inf); Xyzz s3; add(s3, inf, acc); xyzz_to_ref_affine(s3, got); xyzz_to_ref_affine(acc, g2); CHECK(!memcmp(got, g2, 96), "inf+P");
      add(s3, acc, inf); xyzz_to_ref_affine(s3, got); CHECK(!memcmp(got, g2, 96), "P+inf"); add(s3, inf, inf); CHECK(is_infinity(s3), "inf+inf");
      dbl(s3, inf); CHECK(is_infinity(s3), "dbl inf");
      // xyzz store/load roundtrip
      uint32_t w[32]; store_xyzz(w, acc); Xyzz l; load_xyzz(l, w); xyzz_to_ref_affine(l, got); CHECK(!memcmp(got, g2, 96), "xyzz store/load"); }
    // the accumulation's hot-loop form: rows in the device format (packed Montgomery-261), sign applied on the packed words
    // (load_affine_m261_signed), flagged in-place addition (madd_ip) -- the same chain as above plus P + P, P - P and a restart
    { Xyzz hacc; set_infinity(hacc); bool hinf = true;
      uint64_t hr[12]; memset(hr, 0, 96); hr[7] = 1ULL << 63;
      auto step = [&](int i, bool negf) {
          AffineV<2> a; load_ref_affine(a, &srs[8 * i]); uint32_t w[16]; store_affine_m261(w, a.x, a.y);
          Fe<Fq, 1, 1> px; Fe<Fq, 1, 2> py; load_affine_m261_signed(px, py, w, negf);
          for (int l = 0; l < NL - 1; l++) CHECK(px.d[l] < (1u << 29) && py.d[l] < (1u << 29), "signed loader: exact limbs");
          if (hinf) { hacc.x = px; hacc.y = py; hacc.zz = fe_one<Fq>(); hacc.zzz = fe_one<Fq>(); hinf = false; }
          else madd_ip(hacc, hinf, px, py);
          uint64_t pt[8]; memcpy(pt, &srs[8 * i], 64); if (negf) orc_neg(ORC_FQ, pt + 4, pt + 4);
          orc_g1_mixed_add(hr, pt, hr);
          ref_norm(hr, want);
          if (hinf) { CHECK(is_infinity(hacc), "madd_ip leaves a clean infinity"); CHECK((want[7] >> 63) != 0, "madd_ip infinity flag"); }
          else { xyzz_to_ref_affine(hacc, got); CHECK(!memcmp(got, want, 96), "madd_ip chain"); }
          for (int l = 0; l < NL - 1; l++) CHECK(hacc.x.d[l] < (1u << 29) + 8 && hacc.y.d[l] < (1u << 29) + 8 && hacc.zz.d[l] < (1u << 29) + 8 && hacc.zzz.d[l] < (1u << 29) + 8, "madd_ip limb bounds");
      };
      for (int i = 0; i < N; i++) step(i, (i % 3) == 1);
      step(5, false); step(5, false);                 // ... + P + P: the second hits acc == operand only by accident; force it:
      set_infinity(hacc); hinf = true; memset(hr, 0, 96); hr[7] = 1ULL << 63;
      step(7, false); step(7, false);                 // P + P -> doubling branch
      step(9, true);
      set_infinity(hacc); hinf = true; memset(hr, 0, 96); hr[7] = 1ULL << 63;
      step(11, true); step(11, false);                // -P + P -> infinity, flag set
      CHECK(hinf, "madd_ip P + (-P) sets the flag");
      step(12, false); step(13, true);                // restart from the flag
    }
    printf("group ok (fails so far %d)\n", fails);
}

int main()
{
    test_field<FqP>(ORC_FQ, "fq");
    test_field<FrP>(ORC_FR, "fr");
    test_group();
    printf(fails ? "FAILED %d\n" : "ALL OK %d\n", fails);
    return fails ? 1 : 0;
}
