// plonk_driver.cpp -- TEST INFRASTRUCTURE ONLY (BASELINE config 5).
//
// Our own driver around the reference's UNMODIFIED PLONK stack (StandardComposer -> waffle::Prover -> waffle::Verifier),
// compiled where the reference sources lie by oracle/Makefile into two executables:
//   oracle/_ref/plonk_cpu : everything from the reference (its x86-64 asm MSM / FFT)                      -> golden proof
//   oracle/_ref/plonk_gpu : same objects, but pippenger / batched_scalar_multiplications / the fft family are
//                           localised away (objcopy) and resolved by barretenberg_amd/libbbshim.so -> libbbgpu.so (MI355X)
// Both print the proof (9 affine commitments + 12 field evaluations, waffle_types.hpp:18-45) as hex; a drop-in GPU path
// must print byte-identical output (Fiat-Shamir challenges hash the commitments, challenge.hpp:15-135).
//
//   plonk_xxx transcript <path> <num_points>   write a synthetic SRS in the reference's transcript format (io.hpp:36-182)
//   plonk_xxx prove <num_gates>                build the add/mul-chain circuit of test/benchmarks/bench_plonk.cpp:25-37 with
//                                              fixed witnesses, prove, verify, print
//   plonk_xxx trace <num_gates>                prove + also print the Fiat-Shamir challenges (debugging aid for restatements)
//   plonk_xxx vk <num_gates>                   print the verification key waffle::preprocess() derives from the prover state: SIGMA_1..3 and the
//                                              arithmetic widget's five selector commitments (preprocess.hpp:16-55, arithmetic_widget.cpp:128-157)
//   plonk_xxx dump <num_gates> <path>          write the waffle::Prover INPUT state the composer produced (witness values,
//                                              sigma mappings, selector values) as a flat binary file: the input of the native
//                                              resident prover (bbgpu_plonk_*), so that it proves the very same circuit
//   BB_CIRCUIT=bool in the environment switches every mode to a BoolComposer circuit (arithmetic + bool widget): <num_gates> / 2 pairs of
//   bits a_i, b_i constrained boolean, c_i = a_i b_i (mul gate), d_i = a_i + c_i (add gate)
//   BB_CIRCUIT=mimc: a MiMCComposer circuit (arithmetic + MiMC widget): a chain of <num_gates> - 2 MiMC rounds x <- (x + k + c_i)^7 from fixed
//   witnesses, then an addition gate on the result; the proof then carries w_o_shifted_eval and q_mimc_coefficient_eval as well
//   BB_CIRCUIT=extended selects an ExtendedComposer circuit (arithmetic + sequential + bool widgets): products, pairs of chained additions
//   that the composer folds into one gate with a q_o_next term, boolean constraints; the proof then carries w_o_shifted_eval
//   plonk_gpu adapter <num_gates>              the level-2 integration of INTEGRATION.md: reference composer -> bbgpu_plonk_* resident prover
//                                              -> reference Verifier; prints the proof in the `prove` format (GPU-linked builds only)
//   plonk_gpu_full faults <num_gates> <kind> [first [step]]   the error contract under injected GPU failures (see faults() below)
//   plonk_xxx verify <num_gates> < proof       rebuild the same circuit's Verifier and check a proof given in the `prove` text
//                                              format on stdin (used to verify proofs made by the native GPU prover)
#include <barretenberg/curves/bn254/fq.hpp>
#include <barretenberg/curves/bn254/fr.hpp>
#include <barretenberg/curves/bn254/g1.hpp>
#include <barretenberg/curves/bn254/g2.hpp>
#include <barretenberg/waffle/composer/bool_composer.hpp>
#include <barretenberg/waffle/composer/extended_composer.hpp>
#include <barretenberg/waffle/composer/mimc_composer.hpp>
#include <barretenberg/waffle/composer/standard_composer.hpp>
#include <barretenberg/waffle/proof_system/preprocess.hpp>
#include <barretenberg/waffle/proof_system/prover/prover.hpp>
#include <barretenberg/waffle/proof_system/verifier/verifier.hpp>
#include <barretenberg/waffle/proof_system/widgets/arithmetic_widget.hpp>
#include <barretenberg/waffle/proof_system/widgets/bool_widget.hpp>
#include <barretenberg/waffle/proof_system/widgets/mimc_widget.hpp>
#include <barretenberg/waffle/proof_system/widgets/sequential_widget.hpp>
#include <barretenberg/waffle/stdlib/field/field.hpp>

#include <arpa/inet.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

using namespace barretenberg;

// ---- the level-2 integration of INTEGRATION.md, made executable: the C ABI of libbbgpu.so (include/bbgpu.h), bound weakly so that the
// all-CPU build (plonk_cpu, no GPU library on its link line) still links; `adapter` mode needs one of the GPU-linked builds
extern "C" {
#include "../include/bbgpu.h"
}
#pragma weak bbgpu_srs_register
#pragma weak bbgpu_plonk_prover_create
#pragma weak bbgpu_plonk_construct_proof
#pragma weak bbgpu_plonk_prover_destroy
#pragma weak bbgpu_last_error
// the shim's accounting of the drop-in path (bb_shim.cpp, BBGPU_SHIM_PROFILE): reset before / written after the timed call when linked in
extern "C" void bbshim_profile_reset(void);
extern "C" void bbshim_profile_write(const char* tag, double caller_ms);
#pragma weak bbshim_profile_reset
#pragma weak bbshim_profile_write
// faults mode (below): the shim's count of calls answered on the host, the library's fault injection and accounting
extern "C" unsigned long long bbshim_fallback_calls(void);
#pragma weak bbshim_fallback_calls
#pragma weak bbgpu_fault_inject
#pragma weak bbgpu_fault_stats
#pragma weak bbgpu_shutdown
#pragma weak bbgpu_memory_stats

namespace {
// the synthetic SRS secret (fixed, public: this is a test SRS)
fr::field_t secret()
{
    fr::field_t x = { { 0x0123456789abcdefULL, 0xfedcba9876543210ULL, 0x0f1e2d3c4b5a6978ULL, 0x0123456789abcdefULL } };
    return fr::to_montgomery_form(x);
}
void put_fq(std::vector<unsigned char>& out, const fq::field_t& mont)
{
    fq::field_t v = fq::from_montgomery_form(mont); // file holds non-Montgomery limbs, each big-endian, limb 0 first
    for (int l = 0; l < 4; l++)
        for (int b = 7; b >= 0; b--) out.push_back((unsigned char)(v.data[l] >> (8 * b)));
}
void put_u32(std::vector<unsigned char>& out, uint32_t v)
{
    uint32_t be = htonl(v);
    unsigned char* p = (unsigned char*)&be;
    out.insert(out.end(), p, p + 4);
}

int write_transcript(const char* path, size_t num)
{
    const fr::field_t x = secret();
    std::vector<unsigned char> out;
    // io.hpp:36-46: transcript_number, total_transcripts, total_g1_points, total_g2_points, num_g1_points, num_g2_points, start_from
    put_u32(out, 0); put_u32(out, 1); put_u32(out, (uint32_t)num); put_u32(out, 2); put_u32(out, (uint32_t)num); put_u32(out, 2); put_u32(out, 0);
    g1::affine_element p = g1::group_exponentiation(g1::affine_one(), x); // file point k = x^(k+1) G
    for (size_t k = 0; k < num; k++) {
        put_fq(out, p.x);
        put_fq(out, p.y);
        if (k + 1 < num) p = g1::group_exponentiation(p, x);
    }
    g2::affine_element h = g2::affine_one();
    g2::affine_element hx = g2::group_exponentiation(h, x);
    const g2::affine_element g2pts[2] = { h, hx }; // index 1 must be x * G2 (io.hpp:171-180)
    for (const auto& q : g2pts) {
        put_fq(out, q.x.c0); put_fq(out, q.x.c1); put_fq(out, q.y.c0); put_fq(out, q.y.c1);
    }
    out.insert(out.end(), 64, 0); // checksum slot, never verified by the reference
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    printf("wrote %zu G1 points + 2 G2 points to %s\n", num, path);
    return 0;
}

void hex4(const char* name, const uint64_t* d)
{
    printf("%s %016lx%016lx%016lx%016lx\n", name, d[3], d[2], d[1], d[0]);
}

void build_circuit(waffle::StandardComposer& composer, size_t num_gates);
std::unique_ptr<waffle::ComposerBase> make_circuit(size_t num_gates);

int prove(size_t num_gates, bool trace)
{
    // BB_WARM_PROOFS=k: k untimed proofs of the same circuit first (fresh composer and prover each; a Prover is consumed by its proof),
    // so that the timed one is a steady-state proof -- library initialised, SRS and transform tables resident, clocks up
    if (const char* w = getenv("BB_WARM_PROOFS")) {
        for (int i = 0; i < atoi(w); i++) {
            std::unique_ptr<waffle::ComposerBase> c0 = make_circuit(num_gates);
            waffle::Prover p0 = c0->preprocess();
            (void)p0.construct_proof();
        }
    }
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    waffle::Verifier verifier = waffle::preprocess(prover);
    if (bbshim_profile_reset) bbshim_profile_reset();
    auto t0 = std::chrono::steady_clock::now();
    waffle::plonk_proof proof = prover.construct_proof();
    auto t1 = std::chrono::steady_clock::now();
    if (bbshim_profile_write) bbshim_profile_write("construct_proof", std::chrono::duration<double, std::milli>(t1 - t0).count());
    bool ok = verifier.verify_proof(proof);
    printf("n %zu\n", prover.n);
    const g1::affine_element* pts[9] = { &proof.W_L, &proof.W_R, &proof.W_O, &proof.Z_1, &proof.T_LO, &proof.T_MID, &proof.T_HI, &proof.PI_Z, &proof.PI_Z_OMEGA };
    const char* pn[9] = { "W_L", "W_R", "W_O", "Z_1", "T_LO", "T_MID", "T_HI", "PI_Z", "PI_Z_OMEGA" };
    for (int i = 0; i < 9; i++) {
        char nm[32];
        snprintf(nm, sizeof nm, "%s.x", pn[i]); hex4(nm, pts[i]->x.data);
        snprintf(nm, sizeof nm, "%s.y", pn[i]); hex4(nm, pts[i]->y.data);
    }
    const fr::field_t* ev[12] = { &proof.w_l_eval, &proof.w_r_eval, &proof.w_o_eval, &proof.sigma_1_eval, &proof.sigma_2_eval, &proof.z_1_shifted_eval,
                                  &proof.linear_eval, &proof.w_l_shifted_eval, &proof.w_r_shifted_eval, &proof.w_o_shifted_eval, &proof.q_c_eval,
                                  &proof.q_mimc_coefficient_eval };
    const char* en[12] = { "w_l_eval", "w_r_eval", "w_o_eval", "sigma_1_eval", "sigma_2_eval", "z_1_shifted_eval", "linear_eval", "w_l_shifted_eval",
                           "w_r_shifted_eval", "w_o_shifted_eval", "q_c_eval", "q_mimc_coefficient_eval" };
    for (int i = 0; i < 7; i++) hex4(en[i], ev[i]->data); // the standard arithmetic circuit fills these; the rest stay unset
    if (prover.widgets.size() > 1 && dynamic_cast<const waffle::ProverMiMCWidget*>(prover.widgets[1].get())) {
        hex4(en[9], ev[9]->data);   // w_o_shifted_eval (prover.cpp:499-502)
        hex4(en[11], ev[11]->data); // q_mimc_coefficient_eval (mimc_widget.cpp:92-95)
    }
    if (prover.widgets.size() > 1 && dynamic_cast<const waffle::ProverSequentialWidget*>(prover.widgets[1].get())) hex4(en[9], ev[9]->data);
    if (trace) {
        hex4("beta", prover.challenges.beta.data);
        hex4("gamma", prover.challenges.gamma.data);
        hex4("alpha", prover.challenges.alpha.data);
        hex4("z", prover.challenges.z.data);
        hex4("nu", prover.challenges.nu.data);
    }
    printf("verified %d\n", ok ? 1 : 0);
    fprintf(stderr, "construct_proof %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    return ok ? 0 : 2;
}

void build_circuit(waffle::StandardComposer& composer, size_t num_gates)
{
    fr::field_t a0 = fr::to_montgomery_form({ { 0x1111111122222222ULL, 0x3333333344444444ULL, 0x5555555566666666ULL, 0x0777777788888888ULL } });
    fr::field_t b0 = fr::to_montgomery_form({ { 0x9999aaaabbbbccccULL, 0xddddeeeeffff0000ULL, 0x1234123412341234ULL, 0x0abcdefabcdefabcULL } });
    plonk::stdlib::field_t a(plonk::stdlib::witness_t(&composer, a0));
    plonk::stdlib::field_t b(plonk::stdlib::witness_t(&composer, b0));
    plonk::stdlib::field_t c(&composer);
    for (size_t i = 0; i < (num_gates / 4) - 4; ++i) { // bench_plonk.cpp:30-36
        c = a + b;
        c = a * c;
        a = b * b;
        b = c * c;
    }
}

// BoolComposer circuit (bool_composer.cpp): bits with boolean constraints, products and sums of them
void build_bool_circuit(waffle::BoolComposer& composer, size_t num_gates)
{
    const fr::field_t one = fr::one, zero = fr::zero, minus_one = fr::neg_one();
    for (size_t i = 0; i < num_gates / 2; ++i) {
        const bool abit = (i * 7 + 1) & 1, bbit = ((i * 5 + 3) >> 1) & 1; // pair 0 is (1, 1): no all-zero wire polynomial even at n = 4
        const uint32_t a = composer.add_variable(abit ? one : zero), b = composer.add_variable(bbit ? one : zero);
        const uint32_t c = composer.add_variable((abit && bbit) ? one : zero);
        const uint32_t d = composer.add_variable(fr::add(abit ? one : zero, (abit && bbit) ? one : zero));
        composer.create_bool_gate(a);
        composer.create_bool_gate(b);
        composer.create_mul_gate({ a, b, c, one, minus_one, zero });
        composer.create_add_gate({ a, c, d, one, one, minus_one, zero });
    }
}
// MiMCComposer circuit (mimc_composer.cpp): rounds x_out = (x_in + k + c)^7 as one gate each (w_l = k, w_r = (x_in + k + c)^3, w_o = x_in, next w_o = x_out)
void build_mimc_circuit(waffle::MiMCComposer& composer, size_t num_gates)
{
    fr::field_t x = fr::to_montgomery_form({ { 0x1111111122222222ULL, 0x3333333344444444ULL, 0x5555555566666666ULL, 0x0777777788888888ULL } });
    const fr::field_t k = fr::to_montgomery_form({ { 0x9999aaaabbbbccccULL, 0xddddeeeeffff0000ULL, 0x1234123412341234ULL, 0x0abcdefabcdefabcULL } });
    const uint32_t k_idx = composer.add_variable(k);
    uint32_t x_idx = composer.add_variable(x);
    const uint32_t x0_idx = x_idx;
    for (size_t i = 0; i + 2 < num_gates; ++i) {
        const fr::field_t c = fr::to_montgomery_form({ { 0x1000 + 7 * i, i * i + 3, 5, 0 } });
        const fr::field_t t0 = fr::add(fr::add(x, k), c);
        const fr::field_t cubed = fr::mul(fr::sqr(t0), t0);
        const fr::field_t out = fr::mul(fr::sqr(cubed), t0);
        const uint32_t cubed_idx = composer.add_variable(cubed), out_idx = composer.add_variable(out);
        composer.create_mimc_gate({ x_idx, cubed_idx, k_idx, out_idx, c });
        x = out;
        x_idx = out_idx;
    }
    const uint32_t sum_idx = composer.add_variable(fr::add(x, composer.get_variable(x0_idx)));
    composer.create_add_gate({ x_idx, x0_idx, sum_idx, fr::one, fr::one, fr::neg_one(), fr::zero });
}
// ExtendedComposer circuit (extended_composer.cpp): per block of 4 gates a product, two chained additions whose shared wire appears nowhere
// else (combine_linear_relations() folds them into ONE gate with a q_o_next term, extended_composer.cpp:213-440) and a boolean constraint
void build_extended_circuit(waffle::ExtendedComposer& composer, size_t num_gates)
{
    const fr::field_t one = fr::one, zero = fr::zero, minus_one = fr::neg_one();
    fr::field_t a = fr::to_montgomery_form({ { 0x1111111122222222ULL, 0x3333333344444444ULL, 0x5555555566666666ULL, 0x0777777788888888ULL } });
    fr::field_t b = fr::to_montgomery_form({ { 0x9999aaaabbbbccccULL, 0xddddeeeeffff0000ULL, 0x1234123412341234ULL, 0x0abcdefabcdefabcULL } });
    for (size_t i = 0; i < num_gates / 4; ++i) {
        const fr::field_t bit = ((i * 3 + 1) & 2) ? one : zero;
        const fr::field_t c = fr::mul(a, b), e = fr::add(c, a), g = fr::add(e, bit);
        const uint32_t ai = composer.add_variable(a), bi = composer.add_variable(b), ci = composer.add_variable(c);
        const uint32_t ei = composer.add_variable(e), biti = composer.add_variable(bit), gi = composer.add_variable(g);
        composer.create_mul_gate({ ai, bi, ci, one, minus_one, zero });
        composer.create_add_gate({ ci, ai, ei, one, one, minus_one, zero });
        composer.create_add_gate({ ei, biti, gi, one, one, minus_one, zero });
        composer.create_bool_gate(biti);
        a = fr::add(g, b);
        b = fr::sqr(c);
    }
}
// a circuit whose w_r and w_o are identically zero (a * 0 = 0 in every gate): the wire commitments W_R and W_O are the point at infinity
// -- what does the reference put into the proof, and into the transcript hash, for it?  (tests/golden/infinity_commitments.json)
void build_zero_wire_circuit(waffle::StandardComposer& composer, size_t num_gates)
{
    fr::field_t a = fr::to_montgomery_form({ { 0x1111111122222222ULL, 0x3333333344444444ULL, 0x5555555566666666ULL, 0x0777777788888888ULL } });
    const uint32_t zero_idx = composer.add_variable(fr::zero);
    for (size_t i = 0; i < num_gates; ++i) {
        const uint32_t ai = composer.add_variable(a);
        composer.create_mul_gate({ ai, zero_idx, zero_idx, fr::one, fr::neg_one(), fr::zero });
        a = fr::add(fr::sqr(a), fr::one);
    }
}
std::unique_ptr<waffle::ComposerBase> make_circuit(size_t num_gates)
{
    const char* kind = getenv("BB_CIRCUIT");
    if (kind && !strcmp(kind, "zerowire")) {
        auto c = std::make_unique<waffle::StandardComposer>(num_gates);
        build_zero_wire_circuit(*c, num_gates);
        return c;
    }
    if (kind && !strcmp(kind, "extended")) {
        auto c = std::make_unique<waffle::ExtendedComposer>(num_gates);
        build_extended_circuit(*c, num_gates);
        return c;
    }
    if (kind && !strcmp(kind, "mimc")) {
        auto c = std::make_unique<waffle::MiMCComposer>(num_gates);
        build_mimc_circuit(*c, num_gates);
        return c;
    }
    if (kind && !strcmp(kind, "bool")) {
        auto c = std::make_unique<waffle::BoolComposer>(num_gates);
        build_bool_circuit(*c, num_gates);
        return c;
    }
    auto c = std::make_unique<waffle::StandardComposer>(num_gates);
    build_circuit(*c, num_gates);
    return c;
}

void wr(FILE* f, const void* p, size_t bytes) { if (fwrite(p, 1, bytes, f) != bytes) abort(); }

int dump(size_t num_gates, const char* path)
{
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    const waffle::ProverArithmeticWidget* w = dynamic_cast<const waffle::ProverArithmeticWidget*>(prover.widgets[0].get());
    const waffle::ProverBoolWidget* wb = prover.widgets.size() > 1 ? dynamic_cast<const waffle::ProverBoolWidget*>(prover.widgets[1].get()) : nullptr;
    const waffle::ProverMiMCWidget* wm = prover.widgets.size() > 1 ? dynamic_cast<const waffle::ProverMiMCWidget*>(prover.widgets[1].get()) : nullptr;
    const waffle::ProverSequentialWidget* ws = prover.widgets.size() == 3 ? dynamic_cast<const waffle::ProverSequentialWidget*>(prover.widgets[1].get()) : nullptr;
    if (ws) wb = dynamic_cast<const waffle::ProverBoolWidget*>(prover.widgets[2].get()); // ExtendedComposer: arithmetic, sequential, bool
    if (!w || prover.widgets.size() > 3 || (prover.widgets.size() == 2 && !wb && !wm) || (prover.widgets.size() == 3 && (!ws || !wb))) return 3;
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    const uint64_t n = prover.n;
    wr(f, "BBPLONK1", 8);
    wr(f, &n, 8);
    barretenberg::polynomial* wires[3] = { &prover.w_l, &prover.w_r, &prover.w_o };
    for (auto* p : wires) wr(f, p->get_coefficients(), n * 32);
    const std::vector<uint32_t>* maps[3] = { &prover.sigma_1_mapping, &prover.sigma_2_mapping, &prover.sigma_3_mapping };
    for (auto* m : maps) wr(f, m->data(), n * 4);
    const barretenberg::polynomial* sel[5] = { &w->q_m, &w->q_l, &w->q_r, &w->q_o, &w->q_c };
    for (auto* p : sel) wr(f, const_cast<barretenberg::polynomial*>(p)->get_coefficients(), n * 32);
    if (wb) { // bool widget selectors follow (bool_widget.hpp)
        const barretenberg::polynomial* bsel[3] = { &wb->q_bl, &wb->q_br, &wb->q_bo };
        for (auto* p : bsel) wr(f, const_cast<barretenberg::polynomial*>(p)->get_coefficients(), n * 32);
    }
    if (wm) { // MiMC widget selectors follow (mimc_widget.hpp): selector, then round constants
        const barretenberg::polynomial* msel[2] = { &wm->q_mimc_selector, &wm->q_mimc_coefficient };
        for (auto* p : msel) wr(f, const_cast<barretenberg::polynomial*>(p)->get_coefficients(), n * 32);
    }
    if (ws) wr(f, const_cast<barretenberg::polynomial&>(ws->q_o_next).get_coefficients(), n * 32); // sequential widget selector comes last
    fclose(f);
    printf("n %zu\n", (size_t)n);
    return 0;
}

int vk(size_t num_gates)
{
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    waffle::Verifier verifier = waffle::preprocess(prover);
    printf("n %zu\n", prover.n);
    const g1::affine_element* pts[3] = { &verifier.SIGMA_1, &verifier.SIGMA_2, &verifier.SIGMA_3 };
    const char* pn[3] = { "SIGMA_1", "SIGMA_2", "SIGMA_3" };
    char nm[32];
    for (int i = 0; i < 3; i++) {
        snprintf(nm, sizeof nm, "%s.x", pn[i]); hex4(nm, pts[i]->x.data);
        snprintf(nm, sizeof nm, "%s.y", pn[i]); hex4(nm, pts[i]->y.data);
    }
    const char* qn[5] = { "Q_M", "Q_L", "Q_R", "Q_O", "Q_C" };
    const auto& inst = verifier.verifier_widgets[0]->instance;
    for (int i = 0; i < 5; i++) {
        snprintf(nm, sizeof nm, "%s.x", qn[i]); hex4(nm, inst[i].x.data);
        snprintf(nm, sizeof nm, "%s.y", qn[i]); hex4(nm, inst[i].y.data);
    }
    if (verifier.verifier_widgets.size() == 3) { // ExtendedComposer: sequential widget (sequential_widget.cpp:79-106), then the bool widget
        const char* xn[4] = { "Q_O_NEXT", "Q_BL", "Q_BR", "Q_BO" };
        for (int i = 0; i < 4; i++) {
            const auto& pt = i == 0 ? verifier.verifier_widgets[1]->instance[0] : verifier.verifier_widgets[2]->instance[i - 1];
            snprintf(nm, sizeof nm, "%s.x", xn[i]); hex4(nm, pt.x.data);
            snprintf(nm, sizeof nm, "%s.y", xn[i]); hex4(nm, pt.y.data);
        }
    } else if (verifier.verifier_widgets.size() > 1 && verifier.verifier_widgets[1]->instance.size() == 2) { // MiMC widget (mimc_widget.cpp:133-160)
        const char* mn[2] = { "Q_MIMC_COEFFICIENT", "Q_MIMC_SELECTOR" };
        const auto& minst = verifier.verifier_widgets[1]->instance;
        for (int i = 0; i < 2; i++) {
            snprintf(nm, sizeof nm, "%s.x", mn[i]); hex4(nm, minst[i].x.data);
            snprintf(nm, sizeof nm, "%s.y", mn[i]); hex4(nm, minst[i].y.data);
        }
    } else if (verifier.verifier_widgets.size() > 1) { // bool widget: commitments to q_bl, q_br, q_bo (bool_widget.cpp:118-152)
        const char* bn[3] = { "Q_BL", "Q_BR", "Q_BO" };
        const auto& binst = verifier.verifier_widgets[1]->instance;
        for (int i = 0; i < 3; i++) {
            snprintf(nm, sizeof nm, "%s.x", bn[i]); hex4(nm, binst[i].x.data);
            snprintf(nm, sizeof nm, "%s.y", bn[i]); hex4(nm, binst[i].y.data);
        }
    }
    return 0;
}

// What a barretenberg maintainer adds around Composer::preprocess() / Prover::construct_proof() to prove on the resident GPU prover:
// the reference composer builds the circuit, its Prover state is handed to bbgpu_plonk_prover_create as it is, the proof comes back in
// waffle::plonk_proof's own layout and the reference's own Verifier checks it.
int adapter(size_t num_gates)
{
    if (!bbgpu_plonk_prover_create) {
        fprintf(stderr, "adapter: this build is not linked against libbbgpu.so (use plonk_gpu / plonk_gpu_full)\n");
        return 5;
    }
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    waffle::Verifier verifier = waffle::preprocess(prover);
    const waffle::ProverArithmeticWidget* w = dynamic_cast<const waffle::ProverArithmeticWidget*>(prover.widgets[0].get());
    if (!w) return 3;
    bbgpu_plonk_circuit c;
    memset(&c, 0, sizeof c);
    c.n = prover.n;
    c.w_l = (const uint64_t*)prover.w_l.get_coefficients();
    c.w_r = (const uint64_t*)prover.w_r.get_coefficients();
    c.w_o = (const uint64_t*)prover.w_o.get_coefficients();
    c.sigma_1_mapping = prover.sigma_1_mapping.data();
    c.sigma_2_mapping = prover.sigma_2_mapping.data();
    c.sigma_3_mapping = prover.sigma_3_mapping.data();
    auto co = [](const barretenberg::polynomial& p) { return (const uint64_t*)const_cast<barretenberg::polynomial&>(p).get_coefficients(); };
    c.q_m = co(w->q_m); c.q_l = co(w->q_l); c.q_r = co(w->q_r); c.q_o = co(w->q_o); c.q_c = co(w->q_c);
    for (size_t i = 1; i < prover.widgets.size(); i++) {
        if (auto* wb = dynamic_cast<const waffle::ProverBoolWidget*>(prover.widgets[i].get())) { c.q_bl = co(wb->q_bl); c.q_br = co(wb->q_br); c.q_bo = co(wb->q_bo); }
        else if (auto* wm = dynamic_cast<const waffle::ProverMiMCWidget*>(prover.widgets[i].get())) { c.q_mimc_selector = co(wm->q_mimc_selector); c.q_mimc_coefficient = co(wm->q_mimc_coefficient); }
        else if (auto* ws = dynamic_cast<const waffle::ProverSequentialWidget*>(prover.widgets[i].get())) c.q_o_next = co(ws->q_o_next);
        else return 3;
    }
    const int srs = bbgpu_srs_register((const uint64_t*)prover.reference_string.monomials, prover.n);
    const int h = srs >= 0 ? bbgpu_plonk_prover_create(&c, srs) : -1;
    if (h < 0) {
        fprintf(stderr, "adapter: %s\n", bbgpu_last_error ? bbgpu_last_error() : "bbgpu error");
        return 6;
    }
    uint64_t words[BBGPU_PLONK_PROOF_WORDS];
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) { // the first call also prepares the circuit-only polynomials
        auto t0 = std::chrono::steady_clock::now();
        if (bbgpu_plonk_construct_proof(h, words) != 0) return 6;
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    bbgpu_plonk_prover_destroy(h);
    waffle::plonk_proof proof;
    static_assert(sizeof(g1::affine_element) == 64 && sizeof(fr::field_t) == 32, "layout");
    memcpy(&proof.W_L, words, 9 * 64);            // W_L, W_R, W_O, Z_1, T_LO, T_MID, T_HI, PI_Z, PI_Z_OMEGA (waffle_types.hpp:20-28)
    memcpy(&proof.w_l_eval, words + 72, 12 * 32); // the twelve evaluations in declaration order (:30-43)
    const bool ok = verifier.verify_proof(proof);
    printf("n %zu\n", prover.n);
    const g1::affine_element* pts[9] = { &proof.W_L, &proof.W_R, &proof.W_O, &proof.Z_1, &proof.T_LO, &proof.T_MID, &proof.T_HI, &proof.PI_Z, &proof.PI_Z_OMEGA };
    const char* pn[9] = { "W_L", "W_R", "W_O", "Z_1", "T_LO", "T_MID", "T_HI", "PI_Z", "PI_Z_OMEGA" };
    for (int i = 0; i < 9; i++) {
        char nm[32];
        snprintf(nm, sizeof nm, "%s.x", pn[i]); hex4(nm, pts[i]->x.data);
        snprintf(nm, sizeof nm, "%s.y", pn[i]); hex4(nm, pts[i]->y.data);
    }
    const fr::field_t* ev[7] = { &proof.w_l_eval, &proof.w_r_eval, &proof.w_o_eval, &proof.sigma_1_eval, &proof.sigma_2_eval, &proof.z_1_shifted_eval, &proof.linear_eval };
    const char* en[7] = { "w_l_eval", "w_r_eval", "w_o_eval", "sigma_1_eval", "sigma_2_eval", "z_1_shifted_eval", "linear_eval" };
    for (int i = 0; i < 7; i++) hex4(en[i], ev[i]->data);
    if (c.q_mimc_selector || c.q_o_next) hex4("w_o_shifted_eval", proof.w_o_shifted_eval.data);
    if (c.q_mimc_selector) hex4("q_mimc_coefficient_eval", proof.q_mimc_coefficient_eval.data);
    printf("verified %d\n", ok ? 1 : 0);
    fprintf(stderr, "construct_proof (resident GPU prover behind the reference composer) %.2f ms\n", best);
    return ok ? 0 : 2;
}

bool rd4(const char* want, uint64_t* d)
{
    char name[64], hex[80];
    if (scanf("%63s %79s", name, hex) != 2 || strcmp(name, want) || strlen(hex) != 64) return false;
    for (int l = 0; l < 4; l++) {
        char buf[17];
        memcpy(buf, hex + 16 * (3 - l), 16);
        buf[16] = 0;
        d[l] = strtoull(buf, nullptr, 16);
    }
    return true;
}

int verify(size_t num_gates)
{
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    waffle::Verifier verifier = waffle::preprocess(prover);
    waffle::plonk_proof proof;
    g1::affine_element* pts[9] = { &proof.W_L, &proof.W_R, &proof.W_O, &proof.Z_1, &proof.T_LO, &proof.T_MID, &proof.T_HI, &proof.PI_Z, &proof.PI_Z_OMEGA };
    const char* pn[9] = { "W_L", "W_R", "W_O", "Z_1", "T_LO", "T_MID", "T_HI", "PI_Z", "PI_Z_OMEGA" };
    char nm[32];
    size_t nn = 0;
    if (scanf("n %zu", &nn) != 1 || nn != prover.n) return 4;
    for (int i = 0; i < 9; i++) {
        snprintf(nm, sizeof nm, "%s.x", pn[i]); if (!rd4(nm, pts[i]->x.data)) return 4;
        snprintf(nm, sizeof nm, "%s.y", pn[i]); if (!rd4(nm, pts[i]->y.data)) return 4;
    }
    fr::field_t* ev[7] = { &proof.w_l_eval, &proof.w_r_eval, &proof.w_o_eval, &proof.sigma_1_eval, &proof.sigma_2_eval, &proof.z_1_shifted_eval, &proof.linear_eval };
    const char* en[7] = { "w_l_eval", "w_r_eval", "w_o_eval", "sigma_1_eval", "sigma_2_eval", "z_1_shifted_eval", "linear_eval" };
    for (int i = 0; i < 7; i++) if (!rd4(en[i], ev[i]->data)) return 4;
    if (prover.widgets.size() > 1 && dynamic_cast<const waffle::ProverMiMCWidget*>(prover.widgets[1].get())) {
        if (!rd4("w_o_shifted_eval", proof.w_o_shifted_eval.data) || !rd4("q_mimc_coefficient_eval", proof.q_mimc_coefficient_eval.data)) return 4;
    }
    if (prover.widgets.size() > 1 && dynamic_cast<const waffle::ProverSequentialWidget*>(prover.widgets[1].get())) {
        if (!rd4("w_o_shifted_eval", proof.w_o_shifted_eval.data)) return 4;
    }
    bool ok = verifier.verify_proof(proof);
    printf("verified %d\n", ok ? 1 : 0);
    return ok ? 0 : 2;
}

// ---- the error contract of the drop-in boundary under INJECTED GPU failures (GPU-linked builds, no BBGPU_SHIM_STRICT) -------------------------
// plonk_gpu_full faults <num_gates> <kind> [first [step]]     kind = alloc | h2d | d2h | launch (include/bbgpu.h, bbgpu_fault_inject)
// Prints the healthy proof in the `prove` format, then counts how often a COLD proof (library shut down before it: every table, workspace and
// staging buffer is allocated again) passes the funnel of <kind> -- N -- and, for k = first, first + step, ... < N: shuts the library down, arms
// "<kind>:k", proves (the k-th allocation / copy / launch check of that proof fails; the shim must answer on the host and carry on), proves once
// more with nothing armed, and prints one line
//   fault <kind> <k> fired F absorbed A fallbacks_failed_proof X fallbacks_next_proof Y proof_same P next_same Q verified V pending S mem_same M
// mem_same compares bbgpu_memory_stats() after the second proof with the same point of the healthy run; the last line is the library's live
// device allocations after a final bbgpu_shutdown() (0 unless an error path leaked).  A d2h failure inside an in-place call is BBGPU_ERR_LOST and
// aborts by design (bb_shim.cpp): `d2h` sweeps are for seeing exactly that from the outside.
struct ProofBytes {
    unsigned char b[9 * 64 + 12 * 32];
    bool ok;
};
ProofBytes prove_once(size_t num_gates)
{
    std::unique_ptr<waffle::ComposerBase> composer = make_circuit(num_gates);
    waffle::Prover prover = composer->preprocess();
    waffle::Verifier verifier = waffle::preprocess(prover);
    waffle::plonk_proof proof = prover.construct_proof();
    ProofBytes r;
    memset(&r, 0, sizeof r);
    memcpy(r.b, &proof.W_L, 9 * 64);
    memcpy(r.b + 9 * 64, &proof.w_l_eval, 7 * 32); // the evaluations the standard arithmetic circuit fills (waffle_types.hpp:30-36)
    r.ok = verifier.verify_proof(proof);
    return r;
}
int faults(size_t num_gates, const char* kind, size_t first, size_t step)
{
    if (!bbgpu_fault_inject || !bbgpu_fault_stats || !bbshim_fallback_calls || !bbgpu_shutdown || !bbgpu_memory_stats) {
        fprintf(stderr, "faults: this build is not linked against libbbgpu.so / libbbshim.so\n");
        return 5;
    }
    if (int rc = prove(num_gates, false)) return rc; // the healthy proof, in full: the caller compares it with the golden proof
    const ProofBytes good = prove_once(num_gates);
    char spec[64];
    // healthy cold run: funnel count of this kind in one proof, memory after two proofs
    bbgpu_shutdown();
    snprintf(spec, sizeof spec, "%s:%llu", kind, ~0ULL >> 1);
    if (bbgpu_fault_inject(spec)) return 6;
    (void)prove_once(num_gates);
    bbgpu_fault_info fi;
    bbgpu_fault_stats(&fi);
    const uint64_t N = !strcmp(kind, "alloc") ? fi.alloc_calls : !strcmp(kind, "h2d") ? fi.h2d_calls : !strcmp(kind, "d2h") ? fi.d2h_calls : fi.launch_checks;
    (void)prove_once(num_gates);
    bbgpu_memory_info m0;
    bbgpu_memory_stats(&m0);
    printf("sites %s %llu\n", kind, (unsigned long long)N);
    for (uint64_t k = first; k < N; k += step ? step : 1) {
        bbgpu_shutdown();
        snprintf(spec, sizeof spec, "%s:%llu", kind, (unsigned long long)k);
        bbgpu_fault_inject(spec);
        const unsigned long long f0 = bbshim_fallback_calls();
        const ProofBytes p1 = prove_once(num_gates);
        const unsigned long long f1 = bbshim_fallback_calls();
        bbgpu_fault_stats(&fi);
        const ProofBytes p2 = prove_once(num_gates);
        const unsigned long long f2 = bbshim_fallback_calls();
        bbgpu_fault_info fj;
        bbgpu_fault_stats(&fj);
        bbgpu_memory_info m1;
        bbgpu_memory_stats(&m1);
        // the point-table cache is keyed by the CALLER's addresses: every proof builds a new ReferenceString, and whether its monomials land where the last
        // proof's were (one cached table) or elsewhere (a second one, the first left to the LRU) depends on what the heap did in between -- a host answer
        // allocates -- so the three srs_* figures are not comparable between runs; they are bounded by BBGPU_SRS_CACHE_BYTES, and a leaked table would
        // still show in the last line (live allocations after shutdown).  Everything else must be what two healthy proofs leave behind.
        m1.srs_points_bytes = m0.srs_points_bytes; m1.srs_table_bytes = m0.srs_table_bytes; m1.srs_auto_bytes = m0.srs_auto_bytes;
        if (fi.absorbed) m1.msm_workspace_bytes = m0.msm_workspace_bytes; // an SRS that lost its window tables to the failure: its MSMs take one bucket set per window (a larger workspace)
        printf("fault %s %llu fired %llu absorbed %llu fallbacks_failed_proof %llu fallbacks_next_proof %llu proof_same %d next_same %d verified %d pending %llu mem_same %d\n", kind,
               (unsigned long long)k, (unsigned long long)fi.fired, (unsigned long long)fi.absorbed, f1 - f0, f2 - f1, !memcmp(p1.b, good.b, sizeof good.b),
               !memcmp(p2.b, good.b, sizeof good.b), (int)(p1.ok && p2.ok), (unsigned long long)fj.slots_pending, !memcmp(&m0, &m1, sizeof m0));
        fflush(stdout);
        if (memcmp(&m0, &m1, sizeof m0)) {
            const uint64_t *a = (const uint64_t*)&m0, *b = (const uint64_t*)&m1;
            fprintf(stderr, "faults %s:%llu: bbgpu_memory_stats healthy / after:", kind, (unsigned long long)k);
            for (size_t i = 0; i < sizeof m0 / 8; i++) fprintf(stderr, " %llu/%llu", (unsigned long long)a[i], (unsigned long long)b[i]);
            fprintf(stderr, "\n");
        }
    }
    bbgpu_shutdown();
    bbgpu_fault_inject(nullptr);
    bbgpu_fault_stats(&fi);
    printf("live_after_shutdown %llu allocations %llu bytes\n", (unsigned long long)fi.live_allocations, (unsigned long long)fi.live_bytes);
    return 0;
}
} // namespace

int main(int argc, char** argv)
{
    if (argc >= 4 && !strcmp(argv[1], "faults"))
        return faults((size_t)atol(argv[2]), argv[3], argc >= 5 ? (size_t)atol(argv[4]) : 0, argc >= 6 ? (size_t)atol(argv[5]) : 1);
    if (argc >= 4 && !strcmp(argv[1], "transcript")) return write_transcript(argv[2], (size_t)atol(argv[3]));
    if (argc >= 3 && !strcmp(argv[1], "prove")) return prove((size_t)atol(argv[2]), false);
    if (argc >= 3 && !strcmp(argv[1], "trace")) return prove((size_t)atol(argv[2]), true);
    if (argc >= 3 && !strcmp(argv[1], "vk")) return vk((size_t)atol(argv[2]));
    if (argc >= 4 && !strcmp(argv[1], "dump")) return dump((size_t)atol(argv[2]), argv[3]);
    if (argc >= 3 && !strcmp(argv[1], "verify")) return verify((size_t)atol(argv[2]));
    if (argc >= 3 && !strcmp(argv[1], "adapter")) return adapter((size_t)atol(argv[2]));
    fprintf(stderr, "usage: %s transcript <path> <num_points> | prove|trace|verify|vk <num_gates> | dump <num_gates> <path>\n", argv[0]);
    return 64;
}
