// Host-only members of the shim (barretenberg_amd/shim/bb_shim.cpp): the translation units' remaining externs that the PLONK stack
// never calls (get_optimal_bucket_width, scale_by_generator, compute_multiplicative_subgroup, add, mul, fft_inner_serial).
// Prints their outputs on seeded inputs as hex; tests/test_host_boundary.py compares with the oracle.  Links libbbshim.so; no GPU.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../barretenberg_amd/shim/bb_abi.hpp"
#include "../../barretenberg_amd/csrc/host_fr.hpp"

using namespace barretenberg;
using bbgpu::host::Fr;

static uint64_t sm_state;
static uint64_t splitmix()
{
    uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void fill(std::vector<fr::field_t>& v, uint64_t seed)
{
    sm_state = seed;
    for (auto& e : v) {
        for (int i = 0; i < 4; i++) e.data[i] = splitmix();
        e.data[3] &= 0x3fffffffffffffffULL; // any representative below 2^254 + ... : includes values in [r, 2r)
    }
}
static void dump(const char* tag, const fr::field_t* v, size_t n)
{
    for (size_t i = 0; i < n; i++) std::printf("%s %zu %016lx %016lx %016lx %016lx\n", tag, i, v[i].data[0], v[i].data[1], v[i].data[2], v[i].data[3]);
}

int main()
{
    for (size_t n : { 1ul, 2ul, 9ul, 10ul, 34ul, 35ul, 8192ul, 10000ul, 65536ul, 99999ul, 100000ul, 131072ul, 144834ul, 1048576ul, 2139094ul, 14617149ul })
        std::printf("width %zu %zu\n", n, scalar_multiplication::get_optimal_bucket_width(n));
    const size_t lg = 6, n = 1u << lg;
    evaluation_domain dom;
    dom.size = n;
    dom.log2_size = lg;
    dom.num_threads = 1;
    dom.thread_size = n;
    std::vector<fr::field_t> a(n), b(n), r(n), g(2);
    fill(a, 11);
    fill(b, 22);
    fill(g, 33);
    dump("a", a.data(), n);
    dump("b", b.data(), n);
    dump("g", g.data(), 2);
    polynomial_arithmetic::add(a.data(), b.data(), r.data(), dom);
    dump("add", r.data(), n);
    polynomial_arithmetic::mul(a.data(), b.data(), r.data(), dom);
    dump("mul", r.data(), n);
    r = a;
    polynomial_arithmetic::scale_by_generator(r.data(), dom, g[0], g[1]);
    dump("scale", r.data(), n);
    std::vector<fr::field_t> sub(8);
    polynomial_arithmetic::compute_multiplicative_subgroup(3, dom, sub.data());
    dump("subgroup", sub.data(), 8);
    // round-root table of evaluation_domain (evaluation_domain.cpp:57-75): round s holds w_{2m}^j, j < m = 2^(s+1)
    std::vector<std::vector<fr::field_t>> store;
    std::vector<fr::field_t*> table;
    for (size_t s = 0; s + 1 < lg; s++) {
        const size_t m = (size_t)1 << (s + 1);
        store.emplace_back(m);
        const Fr w = bbgpu::host::fr_root_of_unity((int)(s + 2));
        Fr acc = bbgpu::host::fr_one();
        for (size_t j = 0; j < m; j++) {
            std::memcpy(store.back()[j].data, acc.d, 32);
            acc = bbgpu::host::fr_mul(acc, w);
        }
    }
    for (auto& v : store) table.push_back(v.data());
    r = a;
    polynomial_arithmetic::fft_inner_serial(r.data(), n, table);
    dump("fft_serial", r.data(), n);
    return 0;
}
