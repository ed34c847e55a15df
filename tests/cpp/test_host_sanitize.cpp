// The host-side PRODUCT code of the library (csrc/host_small.hpp, host_g1.hpp, host_g2.hpp, host_fr.hpp, keccak.hpp: the answers to tiny
// sizes, the O(1) tail of an MSM, the transcript's G2 half, the Fiat-Shamir hash) compiled with -fsanitize=address,undefined and driven
// against the oracle (oracle/bn254_oracle.c; test infrastructure).  Built and run by tests/test_host_small.py::test_host_code_under_sanitizers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <mutex>
#include <thread>

#include "../../barretenberg_amd/csrc/host_small.hpp"
#include "../../barretenberg_amd/csrc/host_fallback.hpp"
#include "../../barretenberg_amd/csrc/host_g2.hpp"
#include "../../barretenberg_amd/csrc/keccak.hpp"
#define BBGPU_COPY_POOL_TEST_DELAY 1 // helpers of a (re)started pool get to run before the first job is posted: the window of the round-3 bug
#include "../../barretenberg_amd/csrc/host_copy_pool.hpp"
#include "../../oracle/bn254_oracle.h"

using namespace bbgpu::host;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { printf("FAIL %s:%d ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); fails++; } } while (0)

int main()
{
    // ---- msm_small against the oracle's pippenger, n = 0 .. 40, scalars uniform, zero, one repeated value, non-canonical representatives
    const size_t N = 40;
    std::vector<uint64_t> srs(8 * N), table(16 * N), sc(4 * N);
    uint64_t x[4];
    orc_random_scalars(77, 1, x);
    orc_make_srs(x, N, srs.data());
    orc_generate_point_table(srs.data(), table.data(), N);
    const uint64_t* r_mod = orc_const(1, "modulus");
    for (int mode = 0; mode < 4; mode++) {
        orc_random_scalars(100 + mode, N, sc.data());
        if (mode == 1) memset(sc.data(), 0, 32 * N);
        if (mode == 2) for (size_t i = 1; i < N; i++) memcpy(&sc[4 * i], &sc[0], 32);
        if (mode == 3) { // [r, 2r): what the prover may hand over
            for (size_t i = 0; i < N; i++) {
                unsigned __int128 c = 0;
                for (int l = 0; l < 4; l++) { c += (unsigned __int128)sc[4 * i + l] + r_mod[l]; sc[4 * i + l] = (uint64_t)c; c >>= 64; }
            }
        }
        for (size_t n = 0; n <= N; n += (n < 8 ? 1 : 7)) {
            uint64_t want[12], got[12];
            orc_pippenger(sc.data(), table.data(), n, 0, want);
            uint64_t wn[12];
            orc_g1_normalize(want, wn);
            Xyzz p = msm_small(sc.data(), table.data(), n);
            g1_to_normalised(p, got);
            const bool winf = orc_g1_is_infinity(wn) != 0, ginf = (got[7] >> 63) != 0;
            CHECK(winf == ginf, "msm_small infinity mode %d n %zu", mode, n);
            if (!winf && !ginf) CHECK(!memcmp(wn, got, 64), "msm_small mode %d n %zu", mode, n);
        }
    }
    // ---- divsteps inversion = Fermat inversion, both fields: edge values and 20,000 random canonical values each
    {
        uint64_t st = 0x9e3779b97f4a7c15ULL;
        auto next = [&]() { return orc_splitmix64(&st); };
        const uint64_t* pq = orc_const(0, "modulus");
        int bad_q = 0, bad_r = 0;
        for (int i = 0; i < 20040; i++) {
            Fq a; Fr b;
            for (int l = 0; l < 4; l++) { a.d[l] = next(); b.d[l] = next(); }
            a.d[3] &= 0x1fffffffffffffffULL; b.d[3] &= 0x1fffffffffffffffULL; // < 2^253 < p, r
            if (i < 40) { // small values, p - small, powers of two
                memset(a.d, 0, 32); memset(b.d, 0, 32);
                if (i < 10) { a.d[0] = (uint64_t)i; b.d[0] = (uint64_t)i; }
                else if (i < 20) { memcpy(a.d, pq, 32); a.d[0] -= (uint64_t)(i - 9); memcpy(b.d, r_mod, 32); b.d[0] -= (uint64_t)(i - 9); }
                else { a.d[(i - 20) / 5] = 1ULL << (13 * ((i - 20) % 5)); b.d[(i - 20) / 5] = 1ULL << (12 * ((i - 20) % 5) + 1); }
            }
            const Fq ia = fq_inv(a), fa = fq_inv_fermat(a);
            const Fr ib = fr_inv(b), fb = fr_inv_fermat(b);
            if (memcmp(ia.d, fa.d, 32)) bad_q++;
            if (memcmp(ib.d, fb.d, 32)) bad_r++;
        }
        CHECK(bad_q == 0 && bad_r == 0, "divsteps inversion differs from Fermat: fq %d fr %d of 20040", bad_q, bad_r);
    }
    // ---- batch normalisation = one by one
    {
        Xyzz pts[8];
        uint64_t one_by_one[8][12], batch[8 * 12];
        for (int i = 0; i < 8; i++) pts[i] = i == 3 ? g1_infinity() : msm_small(sc.data() + 4 * i, table.data() + 16 * i, 2);
        for (int i = 0; i < 8; i++) g1_to_normalised(pts[i], one_by_one[i]);
        g1_batch_to_normalised(pts, 8, batch);
        for (int i = 0; i < 8; i++) CHECK(!memcmp(one_by_one[i], batch + 12 * i, 96), "batch normalise %d", i);
    }
    // ---- ntt_small against the oracle, 2 .. 64 elements, all seven kinds, non-canonical inputs
    {
        uint64_t cst[4];
        orc_random_scalars(5, 1, cst);
        for (int lg = 1; lg <= 6; lg++) {
            const size_t n = (size_t)1 << lg;
            for (int kind = 0; kind < 7; kind++) {
                std::vector<uint64_t> a(4 * n), b;
                orc_random_scalars(200 + lg, n, a.data());
                b = a;
                CHECK(orc_ntt(a.data(), n, kind, cst) == 0, "oracle ntt");
                ntt_small(b.data(), lg, kind, cst);
                CHECK(!memcmp(a.data(), b.data(), 32 * n), "ntt_small lg %d kind %d", lg, kind);
            }
        }
    }
    // ---- G2: (k1 + k2) G2 computed directly and by two scalar multiplications of the generator followed by a mixed addition
    {
        Fr k1, k2;
        orc_random_scalars(9, 1, k1.d);
        orc_random_scalars(10, 1, k2.d);
        const Fr k12 = fr_add(k1, k2);
        G2Affine a, b, c;
        CHECK(g2_scalar_mul_affine(G2_ONE, k1, &a) && g2_scalar_mul_affine(G2_ONE, k2, &b) && g2_scalar_mul_affine(G2_ONE, k12, &c), "g2 scalar mul");
        G2Jac ja = { a.x, a.y, { FQ_ONE, { { 0, 0, 0, 0 } } } };
        const G2Jac s = g2_madd(ja, b);
        const Fq2 zi = fq2_inv(s.z), zi2 = fq2_sqr(zi);
        const Fq2 sx = fq2_mul(s.x, zi2), sy = fq2_mul(s.y, fq2_mul(zi2, zi));
        CHECK(!memcmp(&sx, &c.x, 64) && !memcmp(&sy, &c.y, 64), "g2 additivity");
    }
    // ---- Keccak-256 of the empty string and of "abc" (FIPS 202 predecessor, the Ethereum variant the reference uses)
    {
        uint64_t h[4];
        keccak256(nullptr, 0, h);
        const uint8_t want0[32] = { 0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e, 0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0,
                                    0xe5, 0x00, 0xb6, 0x53, 0xca, 0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70 };
        uint8_t got[32];
        memcpy(got, h, 32);
        bool same = !memcmp(got, want0, 32);
        if (!same) { // the digest may be kept as four big-endian words: compare as a multiset of bytes in either order
            uint8_t rev[32];
            for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) rev[8 * i + j] = got[8 * i + 7 - j];
            same = !memcmp(rev, want0, 32);
            if (!same) { for (int i = 0; i < 32; i++) rev[i] = got[31 - i]; same = !memcmp(rev, want0, 32); }
        }
        CHECK(same, "keccak256(\"\")");
        const uint8_t abc[3] = { 'a', 'b', 'c' };
        keccak256(abc, 3, h); // only exercised under the sanitizers (partial block path)
    }
    // ---- the staging copy pool (host_copy_pool.hpp): copies split over helper threads, across shutdown() / restart cycles with the buffers of
    //      the previous cycle FREED in between -- a helper of a restarted pool must not pick up the last job of the previous one (round 3: it
    //      did, comparing the job counter with zero, and copied into a freed staging buffer)
    {
        CopyPool pool;
        for (int cycle = 0; cycle < 4; cycle++) {
            for (int rep = 0; rep < 6; rep++) {
                const size_t bytes = ((size_t)1 << 20) * (1 + rep % 4) + 4096 * rep + (rep & 1 ? 123 : 0);
                std::vector<unsigned char> src(bytes), dst(bytes, 0);
                for (size_t i = 0; i < bytes; i += 61) src[i] = (unsigned char)(i * 31 + cycle + rep);
                pool.copy(dst.data(), src.data(), bytes);
                CHECK(!memcmp(dst.data(), src.data(), bytes), "copy pool: cycle %d rep %d", cycle, rep);
            } // src / dst freed here
            pool.shutdown();
        }
        unsigned char small_src[100], small_dst[100];
        memset(small_src, 7, sizeof small_src);
        pool.copy(small_dst, small_src, sizeof small_src); // below the parallel threshold: the caller's thread alone
        CHECK(!memcmp(small_dst, small_src, sizeof small_src), "copy pool: small copy");
    }
    // ---- round 5: range jobs (CopyPool::for_range -- the full content check of a cached point table) interleaved with copies on one pool
    {
        CopyPool pool;
        struct Ctx { const uint32_t* v; std::atomic<unsigned long long> sum; } ctx{ nullptr, { 0 } };
        for (int rep = 0; rep < 8; rep++) {
            const size_t items = 100000 + 977 * (size_t)rep;
            std::vector<uint32_t> v(items);
            unsigned long long want = 0;
            for (size_t i = 0; i < items; i++) { v[i] = (uint32_t)(i * 2654435761u + (unsigned)rep); want += v[i]; }
            ctx.v = v.data();
            ctx.sum = 0;
            pool.for_range(items, 1024, [](void* c, size_t lo, size_t hi) {
                Ctx* x = static_cast<Ctx*>(c);
                unsigned long long s = 0;
                for (size_t i = lo; i < hi; i++) s += x->v[i];
                x->sum += s;
            }, &ctx);
            CHECK(ctx.sum.load() == want, "copy pool range job %d", rep);
            std::vector<unsigned char> src((size_t)1 << 20, (unsigned char)rep), dst((size_t)1 << 20, 0);
            pool.copy(dst.data(), src.data(), src.size()); // a copy right behind a range job: the helpers must not run the stale function
            CHECK(!memcmp(dst.data(), src.data(), src.size()), "copy after a range job %d", rep);
            if (rep == 4) pool.shutdown();
        }
        // background jobs (post_range / join): the helpers alone work while the caller does something else; a copy issued meanwhile joins the job first
        for (int rep = 0; rep < 6; rep++) {
            const size_t items = 50000 + 1234 * (size_t)rep;
            std::vector<uint32_t> v(items);
            unsigned long long want = 0;
            for (size_t i = 0; i < items; i++) { v[i] = (uint32_t)(i * 40503u + (unsigned)rep); want += v[i]; }
            ctx.v = v.data();
            ctx.sum = 0;
            pool.post_range(items, [](void* c, size_t lo, size_t hi) {
                Ctx* x = static_cast<Ctx*>(c);
                unsigned long long s = 0;
                for (size_t i = lo; i < hi; i++) s += x->v[i];
                x->sum += s;
            }, &ctx);
            if (rep & 1) { // a copy while the job is posted
                std::vector<unsigned char> src((size_t)1 << 20, (unsigned char)(rep + 3)), dst((size_t)1 << 20, 0);
                pool.copy(dst.data(), src.data(), src.size());
                CHECK(!memcmp(dst.data(), src.data(), src.size()), "copy while a background job is posted %d", rep);
            }
            pool.join();
            CHECK(ctx.sum.load() == want, "background range job %d", rep);
            if (rep == 3) pool.shutdown();
        }
        ctx.sum = 0;
        const uint32_t few[3] = { 1, 2, 3 };
        ctx.v = few;
        pool.for_range(3, 1024, [](void* c, size_t lo, size_t hi) { Ctx* x = static_cast<Ctx*>(c); for (size_t i = lo; i < hi; i++) x->sum += x->v[i]; }, &ctx);
        CHECK(ctx.sum.load() == 6, "copy pool: small range job on the caller's thread");
    }
    // ---- round 4: the staging copies are entered from several threads (the resident prover's uploads beside another thread's transform) and take the
    //      library mutex themselves (capi.hip host_to_device / device_to_host_sync); the single-producer pool behind such a guard, two threads at once
    {
        CopyPool pool;
        std::mutex guard; // stands for capi.hip's g_mu
        int bad[2] = { 0, 0 };
        auto worker = [&](int id) {
            for (int rep = 0; rep < 12; rep++) {
                const size_t bytes = ((size_t)1 << 20) + 8192 * (size_t)(rep + id) + (size_t)(id ? 77 : 0);
                std::vector<unsigned char> src(bytes), dst(bytes, 0);
                for (size_t i = 0; i < bytes; i += 53) src[i] = (unsigned char)(i * 7 + (size_t)rep + (size_t)id);
                {
                    std::lock_guard<std::mutex> lk(guard);
                    pool.copy(dst.data(), src.data(), bytes);
                }
                if (memcmp(dst.data(), src.data(), bytes)) bad[id]++;
            }
        };
        std::thread a(worker, 0), b(worker, 1);
        a.join();
        b.join();
        CHECK(bad[0] == 0 && bad[1] == 0, "guarded copy pool from two threads: %d / %d bad copies", bad[0], bad[1]);
    }
    // ---- round 4: the shim's host answers (host_fallback.hpp) against the oracle: MSM at sizes on both sides of its thresholds, the transform family,
    //      evaluate and the synthetic division (threads inside: run under the sanitizers like everything else here)
    {
        const size_t M = 300;
        std::vector<uint64_t> srs2(8 * M), tab2(16 * M), sc2(4 * M);
        orc_make_srs(x, M, srs2.data());
        orc_generate_point_table(srs2.data(), tab2.data(), M);
        orc_random_scalars(321, M, sc2.data());
        for (size_t n : { (size_t)33, (size_t)64, (size_t)255, M }) {
            uint64_t want[12], wn[12], got[12];
            orc_pippenger(sc2.data(), tab2.data(), n, 0, want);
            orc_g1_normalize(want, wn);
            g1_to_normalised(msm_pippenger(sc2.data(), tab2.data(), n, 16), got);
            CHECK(!memcmp(wn, got, 64), "msm_pippenger n %zu", n);
            g1_to_normalised(msm_pippenger(sc2.data(), srs2.data(), n, 8), got);
            CHECK(!memcmp(wn, got, 64), "msm_pippenger (plain table) n %zu", n);
        }
        uint64_t cst[4];
        orc_random_scalars(55, 1, cst);
        for (int lg : { 1, 2, 5, 9 }) {
            const size_t n = (size_t)1 << lg;
            std::vector<uint64_t> in(4 * n), a(4 * n), b(4 * n);
            orc_random_scalars(400 + (uint64_t)lg, n, in.data());
            for (int kind = 0; kind < 7; kind++) {
                a = in;
                b = in;
                orc_ntt(a.data(), n, kind, cst);
                ntt_radix2(b.data(), lg, kind, cst);
                CHECK(a == b, "ntt_radix2 lg %d kind %d", lg, kind);
            }
            uint64_t z[4], want[4];
            orc_random_scalars(77 + (uint64_t)lg, 1, z);
            orc_evaluate(in.data(), z, n, want);
            Fr zz;
            memcpy(zz.d, z, 32);
            const Fr got = poly_evaluate(in.data(), n, zz);
            CHECK(!memcmp(got.d, want, 32), "poly_evaluate lg %d", lg);
            std::vector<uint64_t> q(4 * n);
            const Fr fz = kate_opening(in.data(), q.data(), n, zz);
            CHECK(!memcmp(fz.d, want, 32), "kate_opening F(z) lg %d", lg);
            // (X - z) Q(X) + F(z) = F(X) at another point
            uint64_t y[4], fy[4], qy[4];
            orc_random_scalars(99 + (uint64_t)lg, 1, y);
            orc_evaluate(in.data(), y, n, fy);
            orc_evaluate(q.data(), y, n, qy);
            Fr Y, QY, FY;
            memcpy(Y.d, y, 32); memcpy(QY.d, qy, 32); memcpy(FY.d, fy, 32);
            const Fr lhs = fr_add(fr_mul(fr_sub(Y, fr_mul(zz, fr_one())), QY), fz);
            CHECK(!memcmp(lhs.d, FY.d, 32), "kate_opening identity lg %d", lg);
        }
    }
    printf(fails ? "FAILED %d\n" : "ALL OK %d\n", fails);
    return fails ? 1 : 0;
}
