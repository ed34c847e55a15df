// plonk.hip -- the PLONK prover rounds with every polynomial resident in HBM (BASELINE config 5).
//
// Restates waffle::Prover::construct_proof (src/barretenberg/waffle/proof_system/prover/prover.cpp:661-670) for the
// standard arithmetic circuit (widgets/arithmetic_widget.cpp): the member functions below carry the reference's names and
// cite the lines they follow.  The reference moves every polynomial across the boundary for each of its 26 transforms and
// 9 commitments; here the witness and circuit polynomials are uploaded once, all NTTs (ntt.hip), MSMs (msm.hip) and the
// O(n) loops in between (poly.hip) run on the device, and only 32-byte evaluations and 96-byte commitments come back --
// they have to, because the Fiat-Shamir challenges (challenge.hpp:64-135, Keccak-256 on the host) depend on them.
//
// Bit-exactness: every proof element is a canonical field value or a normalised curve point, both unique, and exact
// arithmetic makes them independent of evaluation order; the proof bytes equal the reference's (tests/golden/plonk_proofs.json).
// State that depends only on the circuit (sigma polynomials, selector transforms) or only on n (subgroup table, L_1 on the
// 2n coset) is computed on first use and kept, where the reference recomputes it inside every construct_proof().
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "bbgpu_internal.h"
#include "host_fr.hpp"
#include "host_g1.hpp"
#include "keccak.hpp"
#include "poly.h"

namespace bbgpu {
namespace {

using host::Fr;

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));                                \
            return BBGPU_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)
#define RC(x)                                                                                                          \
    do {                                                                                                               \
        int rc_ = (x);                                                                                                 \
        if (rc_) return rc_;                                                                                           \
    } while (0)

int ilog2(size_t n)
{
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// waffle_types.hpp:18-45, the part the standard arithmetic circuit fills: 9 commitments (x, y) then 7 evaluations
struct Proof {
    uint64_t W_L[8], W_R[8], W_O[8], Z_1[8], T_LO[8], T_MID[8], T_HI[8], PI_Z[8], PI_Z_OMEGA[8];
    Fr w_l_eval, w_r_eval, w_o_eval, sigma_1_eval, sigma_2_eval, z_1_shifted_eval, linear_eval;
    Fr w_l_shifted_eval, w_r_shifted_eval, w_o_shifted_eval, q_c_eval, q_mimc_coefficient_eval; // widget-dependent (waffle_types.hpp:39-43)
};
static_assert(sizeof(Proof) == BBGPU_PLONK_PROOF_WORDS * 8, "proof layout");

struct Challenges {
    Fr beta, gamma, alpha, z, nu;
};

class PlonkProver {
  public:
    size_t n = 0;
    int log2n = 0;
    int srs = -1;
    hipStream_t st = nullptr;
    hipStream_t st_msm = nullptr;   // the commitments' own queue, ordered after `st` by an event instead of a host round trip
    hipEvent_t scalars_ready = nullptr;
    poly::Scratch scratch;
    std::vector<void*> allocs;

    // inputs (prover.hpp:44-59, arithmetic_widget.hpp:45-49): Lagrange-base values / permutation mappings
    uint64_t* w_lagrange[3] = {};
    uint32_t* sigma_mapping[3] = {};
    uint64_t* q_lagrange[5] = {};
    // circuit- / domain-only state, built on first use
    bool circuit_ready = false;
    uint64_t* roots = nullptr;        // w^i, i < n
    uint64_t* sigma_lagrange[3] = {}; // permutation.hpp:15-87
    uint64_t* q_coeff[5] = {};        // selector polynomials, coefficient form
    uint64_t* q_fft2n[5] = {};        // their coset evaluations on the 2n domain (unscaled)
    uint64_t* l_1 = nullptr;          // L_1 on the 2n coset
    // optional bool widget (bool_widget.hpp): q_bl, q_br, q_bo
    bool has_bool = false;
    uint64_t* qb_lagrange[3] = {};
    uint64_t* qb_coeff[3] = {};
    uint64_t* qb_fft2n[3] = {};
    // optional MiMC widget (mimc_widget.hpp): [0] = q_mimc_selector, [1] = q_mimc_coefficient
    bool has_mimc = false;
    uint64_t* qm_lagrange[2] = {};
    uint64_t* qm_coeff[2] = {};
    uint64_t* qm_fft4n[2] = {};
    // optional sequential widget (sequential_widget.hpp): q_o_next
    bool has_seq = false;
    uint64_t *qs_lagrange = nullptr, *qs_coeff = nullptr, *qs_fft2n = nullptr;
    // per proof
    uint64_t* w[3] = {};        // wire polynomials, coefficient form          (Prover::w_l, w_r, w_o after :130-132)
    uint64_t* sigma[3] = {};    // beta * sigma_i, coefficient form             (after :245-247)
    uint64_t* z = nullptr;      // grand product polynomial, coefficient form   (after :221)
    uint64_t* w_fft[3] = {};    // 4n coset evaluations                          (circuit_state.w_*_fft)
    uint64_t* s_fft[3] = {};    // w_i + beta sigma_i + gamma on the 4n coset
    uint64_t* z_fft = nullptr;  // alpha * Z on the 4n coset
    uint64_t* quotient_large = nullptr; // 4n
    uint64_t* quotient_mid = nullptr;   // 2n
    uint64_t* r = nullptr;              // linearisation polynomial
    uint64_t* tmp[3] = {};              // n-sized workspaces (num / den / opening polynomials)
    uint64_t* slots = nullptr;          // 16 x 32 bytes of device results
    void* h_slots = nullptr;            // pinned mirror

    Challenges challenges;
    Proof proof;
    double timing[8] = {}; // total, msm, ntt+pointwise (the rest), first-use preparation

    ~PlonkProver() { release(); }
    void release()
    {
        if (st) (void)hipStreamSynchronize(st);
        for (void* p : allocs) (void)dev_free(p);
        allocs.clear();
        if (h_slots) (void)hipHostFree(h_slots);
        h_slots = nullptr;
        scratch.release();
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        if (st_msm) (void)hipStreamDestroy(st_msm);
        st_msm = nullptr;
        if (scalars_ready) (void)hipEventDestroy(scalars_ready);
        scalars_ready = nullptr;
    }
    template <class T> int dalloc(T** p, size_t bytes)
    {
        HIPCHK(dev_malloc((void**)p, bytes));
        allocs.push_back(*p);
        return BBGPU_OK;
    }

    int init(const bbgpu_plonk_circuit* c, int srs_handle)
    {
        n = c->n;
        log2n = ilog2(n);
        srs = srs_handle;
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&st_msm, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&scalars_ready, hipEventDisableTiming));
        const size_t fb = n * 32;
        const uint64_t* hw[3] = { c->w_l, c->w_r, c->w_o };
        const uint32_t* hm[3] = { c->sigma_1_mapping, c->sigma_2_mapping, c->sigma_3_mapping };
        const uint64_t* hq[5] = { c->q_m, c->q_l, c->q_r, c->q_o, c->q_c };
        // the three wire / sigma polynomials (and the five selectors) sit back to back so that their transforms run as one
        // batched launch (bbgpu_ntt_device_batch): at these sizes a single transform leaves most of the chip idle
        RC(dalloc(&w_lagrange[0], 3 * fb));
        RC(dalloc(&sigma_lagrange[0], 3 * fb));
        RC(dalloc(&w[0], 3 * fb));
        RC(dalloc(&sigma[0], 3 * fb));
        RC(dalloc(&w_fft[0], 12 * fb));
        RC(dalloc(&s_fft[0], 12 * fb));
        for (int k = 0; k < 3; k++) {
            w_lagrange[k] = w_lagrange[0] + (size_t)k * n * 4;
            sigma_lagrange[k] = sigma_lagrange[0] + (size_t)k * n * 4;
            w[k] = w[0] + (size_t)k * n * 4;
            sigma[k] = sigma[0] + (size_t)k * n * 4;
            w_fft[k] = w_fft[0] + (size_t)k * 4 * n * 4;
            s_fft[k] = s_fft[0] + (size_t)k * 4 * n * 4;
            RC(dalloc(&sigma_mapping[k], n * 4));
            RC(dalloc(&tmp[k], fb));
            RC(host_to_device(w_lagrange[k], hw[k], fb, st));
            RC(host_to_device(sigma_mapping[k], hm[k], n * 4, st));
        }
        RC(dalloc(&q_lagrange[0], 5 * fb));
        RC(dalloc(&q_coeff[0], 5 * fb));
        RC(dalloc(&q_fft2n[0], 10 * fb));
        for (int k = 0; k < 5; k++) {
            q_lagrange[k] = q_lagrange[0] + (size_t)k * n * 4;
            q_coeff[k] = q_coeff[0] + (size_t)k * n * 4;
            q_fft2n[k] = q_fft2n[0] + (size_t)k * 2 * n * 4;
            RC(host_to_device(q_lagrange[k], hq[k], fb, st));
        }
        const uint64_t* hb[3] = { c->q_bl, c->q_br, c->q_bo };
        has_bool = hb[0] != nullptr;
        if (has_bool) {
            RC(dalloc(&qb_lagrange[0], 3 * fb));
            RC(dalloc(&qb_coeff[0], 3 * fb));
            RC(dalloc(&qb_fft2n[0], 6 * fb));
            for (int k = 0; k < 3; k++) {
                qb_lagrange[k] = qb_lagrange[0] + (size_t)k * n * 4;
                qb_coeff[k] = qb_coeff[0] + (size_t)k * n * 4;
                qb_fft2n[k] = qb_fft2n[0] + (size_t)k * 2 * n * 4;
                RC(host_to_device(qb_lagrange[k], hb[k], fb, st));
            }
        }
        const uint64_t* hmm[2] = { c->q_mimc_selector, c->q_mimc_coefficient };
        has_mimc = hmm[0] != nullptr;
        if (has_mimc) {
            RC(dalloc(&qm_lagrange[0], 2 * fb));
            RC(dalloc(&qm_coeff[0], 2 * fb));
            RC(dalloc(&qm_fft4n[0], 8 * fb));
            for (int k = 0; k < 2; k++) {
                qm_lagrange[k] = qm_lagrange[0] + (size_t)k * n * 4;
                qm_coeff[k] = qm_coeff[0] + (size_t)k * n * 4;
                qm_fft4n[k] = qm_fft4n[0] + (size_t)k * 4 * n * 4;
                RC(host_to_device(qm_lagrange[k], hmm[k], fb, st));
            }
        }
        has_seq = c->q_o_next != nullptr;
        if (has_seq) {
            RC(dalloc(&qs_lagrange, fb));
            RC(dalloc(&qs_coeff, fb));
            RC(dalloc(&qs_fft2n, 2 * fb));
            RC(host_to_device(qs_lagrange, c->q_o_next, fb, st));
        }
        RC(dalloc(&roots, fb));
        RC(dalloc(&l_1, 2 * fb));
        RC(dalloc(&z, fb));
        RC(dalloc(&z_fft, 4 * fb));
        RC(dalloc(&quotient_large, 4 * fb));
        RC(dalloc(&quotient_mid, 2 * fb));
        RC(dalloc(&r, fb));
        RC(dalloc(&slots, 16 * 32));
        HIPCHK(hipHostMalloc(&h_slots, 16 * 32));
        HIPCHK(hipStreamSynchronize(st)); // the caller's arrays may go away after this call (uploads above 8 MiB read them asynchronously)
        return BBGPU_OK;
    }
    int set_witness(const uint64_t* wl, const uint64_t* wr, const uint64_t* wo)
    {
        const uint64_t* hw[3] = { wl, wr, wo };
        for (int k = 0; k < 3; k++) RC(host_to_device(w_lagrange[k], hw[k], n * 32, st));
        HIPCHK(hipStreamSynchronize(st));
        return BBGPU_OK;
    }

    // ---- small helpers ----------------------------------------------------------------------------------------------
    int ntt(uint64_t* d, size_t size, int kind, const Fr* c = nullptr) { return bbgpu_ntt_device(d, size, kind, c ? c->d : nullptr, st); }
    // `batch` transforms of consecutive polynomials (the contiguous triples / quintuples above)
    int ntt_batch(uint64_t* d, size_t size, int batch, int kind, const Fr* c = nullptr)
    {
        return bbgpu_ntt_device_batch(d, size, size, batch, kind, c ? c->d : nullptr, st);
    }
    int copy(uint64_t* dst, const uint64_t* src, size_t count)
    {
        HIPCHK(hipMemcpyAsync(dst, src, count * 32, hipMemcpyDeviceToDevice, st));
        return BBGPU_OK;
    }
    // commitments of `count` <= 3 resident coefficient vectors of n scalars: one batched pass (bbgpu_msm_g1_device_batch_async);
    // without window tables on the SRS, side-by-side single MSMs.  Split in two so that work of the NEXT round which does not depend
    // on this round's challenge can be enqueued on our stream in between: a 2^16-point batch is a ~0.4 ms chain of mostly
    // latency-bound launches on the MSM's own queues, beside which transforms run almost for free.
    // Tickets that were issued and never waited on would stay `pending` for the life of the process and turn every later host-pointer MSM
    // into BBGPU_ERR_STATE: whatever happens between commit_begin and commit_end (a failing transform, a failing second ticket), the
    // destructor drains what is still outstanding.
    struct PendingCommit {
        int count = 0, ticket = -1, tk[3] = { -1, -1, -1 };
        bool batched = true;
        double t0 = 0.0;
        PendingCommit() = default;
        PendingCommit(const PendingCommit&) = delete;
        PendingCommit& operator=(const PendingCommit&) = delete;
        ~PendingCommit() { drain(); }
        void drain()
        {
            uint64_t sink[4 * 12];
            if (ticket >= 0) (void)bbgpu_msm_g1_batch_wait(ticket, sink);
            ticket = -1;
            for (int& t : tk) {
                if (t >= 0) (void)bbgpu_msm_g1_wait(t, sink);
                t = -1;
            }
        }
    };
    int commit_begin(const uint64_t* const* scalars, int count, PendingCommit& P)
    {
        P.drain();
        P.batched = true;
        P.t0 = now_ms();
        P.count = count;
        // the scalars are produced on `st`; the commitments run on their own queue behind an event (no host round trip)
        HIPCHK(hipEventRecord(scalars_ready, st));
        HIPCHK(hipStreamWaitEvent(st_msm, scalars_ready, 0));
        const int bt = bbgpu_msm_g1_device_batch_async(srs, 0, scalars, count, n, st_msm);
        if (bt >= 0) {
            P.ticket = bt;
            return BBGPU_OK;
        }
        if (bt != BBGPU_ERR_ARG) return bt;
        P.batched = false;
        const int W = bbgpu_srs_num_windows(srs, n);
        if (W < 0) return W;
        for (int i = 0; i < count; i++) {
            const int t = bbgpu_msm_g1_device_async(srs, 0, scalars[i], n, 0, W, st_msm);
            if (t < 0) return t; // the tickets issued so far are drained by P's destructor
            P.tk[i] = t;
        }
        return BBGPU_OK;
    }
    int commit_end(PendingCommit& P, uint64_t (*out)[8])
    {
        uint64_t res[4 * 12];
        if (P.batched) {
            const int t = P.ticket;
            P.ticket = -1; // a wait consumes the ticket whatever it returns
            RC(bbgpu_msm_g1_batch_wait(t, res));
            for (int i = 0; i < P.count; i++) memcpy(out[i], res + 12 * i, 64); // normalised: x, y canonical
        } else {
            for (int i = 0; i < P.count; i++) {
                const int t = P.tk[i];
                P.tk[i] = -1;
                RC(bbgpu_msm_g1_wait(t, res));
                memcpy(out[i], res, 64);
            }
        }
        timing[1] += now_ms() - P.t0;
        return BBGPU_OK;
    }
    int commit(const uint64_t* const* scalars, int count, uint64_t (*out)[8])
    {
        PendingCommit P;
        RC(commit_begin(scalars, count, P));
        return commit_end(P, out);
    }
    // challenge.hpp:15-62: commitments / evaluations enter the transcript out of Montgomery form
    static void put_point(std::vector<uint64_t>& buf, const uint64_t p[8])
    {
        host::Fq x, y, one = { { 1, 0, 0, 0 } };
        memcpy(x.d, p, 32);
        memcpy(y.d, p + 4, 32);
        x = host::fq_mul(x, one);
        y = host::fq_mul(y, one);
        buf.insert(buf.end(), x.d, x.d + 4);
        buf.insert(buf.end(), y.d, y.d + 4);
    }
    static void put_fr(std::vector<uint64_t>& buf, const Fr& v)
    {
        Fr p = host::fr_from_mont(v);
        buf.insert(buf.end(), p.d, p.d + 4);
    }
    static Fr challenge(const std::vector<uint64_t>& buf)
    {
        Fr h;
        host::hash_field_elements(buf.data(), buf.size() / 4, h.d);
        return host::fr_to_mont(h); // challenge.hpp:70-71: the raw 256-bit digest, reduced by the Montgomery conversion
    }
    std::vector<uint64_t> transcript_upto(int stage) const
    {
        std::vector<uint64_t> b;
        put_point(b, proof.W_L); put_point(b, proof.W_R); put_point(b, proof.W_O);                  // add_wire_commitments_to_buffer
        if (stage >= 1) put_point(b, proof.Z_1);                                                     // add_grand_product_commitments_to_buffer
        if (stage >= 2) { put_point(b, proof.T_LO); put_point(b, proof.T_MID); put_point(b, proof.T_HI); } // add_quotient_commitment_to_buffer
        return b;
    }

    // ---- circuit-only state (first proof) -----------------------------------------------------------------------------
    int prepare_circuit()
    {
        if (circuit_ready) return BBGPU_OK;
        const double t0 = now_ms();
        const Fr root = host::fr_root_of_unity(log2n);
        RC(poly::powers(roots, n, root, host::fr_one(), st));
        for (int k = 0; k < 3; k++) RC(poly::sigma_from_mapping(sigma_lagrange[k], sigma_mapping[k], roots, n, st)); // prover.cpp:663-665
        // arithmetic_widget.cpp:68-84 without the alpha scaling (applied in quotient_mid)
        RC(copy(q_coeff[0], q_lagrange[0], 5 * n));
        RC(ntt_batch(q_coeff[0], n, 5, BBGPU_IFFT));
        for (int k = 0; k < 5; k++) RC(poly::copy_pad(q_fft2n[k], q_coeff[k], n, 2 * n, st));
        RC(ntt_batch(q_fft2n[0], 2 * n, 5, BBGPU_COSET_FFT));
        if (has_bool) { // bool_widget.cpp:64-74 without the alpha scalings (applied in quotient_bool)
            RC(copy(qb_coeff[0], qb_lagrange[0], 3 * n));
            RC(ntt_batch(qb_coeff[0], n, 3, BBGPU_IFFT));
            for (int k = 0; k < 3; k++) RC(poly::copy_pad(qb_fft2n[k], qb_coeff[k], n, 2 * n, st));
            RC(ntt_batch(qb_fft2n[0], 2 * n, 3, BBGPU_COSET_FFT));
        }
        if (has_seq) { // sequential_widget.cpp:49-54 without the alpha scaling (applied in quotient_seq)
            RC(copy(qs_coeff, qs_lagrange, n));
            RC(ntt(qs_coeff, n, BBGPU_IFFT));
            RC(poly::copy_pad(qs_fft2n, qs_coeff, n, 2 * n, st));
            RC(ntt(qs_fft2n, 2 * n, BBGPU_COSET_FFT));
        }
        if (has_mimc) { // mimc_widget.cpp:60-67 without the alpha scaling (applied in quotient_mimc)
            RC(copy(qm_coeff[0], qm_lagrange[0], 2 * n));
            RC(ntt_batch(qm_coeff[0], n, 2, BBGPU_IFFT));
            for (int k = 0; k < 2; k++) RC(poly::copy_pad(qm_fft4n[k], qm_coeff[k], n, 4 * n, st));
            RC(ntt_batch(qm_fft4n[0], 4 * n, 2, BBGPU_COSET_FFT));
        }
        RC(poly::lagrange_l1_fft(l_1, quotient_mid, log2n, log2n + 1, scratch, st)); // prover.cpp:350-351 (quotient_mid as workspace)
        HIPCHK(hipStreamSynchronize(st));
        circuit_ready = true;
        timing[3] = now_ms() - t0;
        return BBGPU_OK;
    }

    // prover.cpp:124-133
    int compute_wire_coefficients()
    {
        RC(copy(w[0], w_lagrange[0], 3 * n));
        RC(ntt_batch(w[0], n, 3, BBGPU_IFFT));
        return BBGPU_OK;
    }
    // prover.cpp:65-86
    int compute_wire_commitments()
    {
        const uint64_t* sc[3] = { w[0], w[1], w[2] };
        uint64_t out[3][8];
        PendingCommit P;
        RC(commit_begin(sc, 3, P));
        // beside the commitments: the wires on the 4n coset (prover.cpp:418-425) need no challenge
        for (int k = 0; k < 3; k++) RC(poly::copy_pad(w_fft[k], w[k], n, 4 * n, st));
        RC(ntt_batch(w_fft[0], 4 * n, 3, BBGPU_COSET_FFT));
        RC(commit_end(P, out));
        memcpy(proof.W_L, out[0], 64);
        memcpy(proof.W_R, out[1], 64);
        memcpy(proof.W_O, out[2], 64);
        std::vector<uint64_t> b = transcript_upto(0);
        challenges.gamma = challenge(b); // compute_gamma, challenge.hpp:64-73
        put_fr(b, challenges.gamma);
        challenges.beta = challenge(b);  // compute_beta, :75-85
        return BBGPU_OK;
    }
    // prover.cpp:135-222: Z(w^m) = prod_{i<m} num_i / den_i.  The six serial accumulator chains (:194-202) and the batch
    // inversion (:215) become one exclusive prefix-product scan of the numerators, one inclusive suffix-product scan of the
    // denominators and a single inversion: 1 / prod_{i<m} den_i = (prod_{i>=m} den_i) / prod_i den_i.
    int compute_z_coefficients()
    {
        poly::ZTermsArgs A{};
        A.w_l = (const uint32_t*)w_lagrange[0]; A.w_r = (const uint32_t*)w_lagrange[1]; A.w_o = (const uint32_t*)w_lagrange[2];
        A.s1 = (const uint32_t*)sigma_lagrange[0]; A.s2 = (const uint32_t*)sigma_lagrange[1]; A.s3 = (const uint32_t*)sigma_lagrange[2];
        A.num = (uint32_t*)tmp[0]; A.den = (uint32_t*)tmp[1];
        A.n = (uint32_t)n;
        RC(poly::z_terms(A, host::fr_root_of_unity(log2n), challenges.beta, challenges.gamma, st));
        // PN: exclusive prefix products of num -> tmp[2];  SD: inclusive suffix products of den -> r (free at this point), total -> slot 0
        poly::ScanJob sj[2] = {};
        sj[0].in = tmp[0]; sj[0].out = tmp[2]; sj[0].n = n; sj[0].reverse = false; sj[0].inclusive = false;
        sj[1].in = tmp[1]; sj[1].out = r; sj[1].n = n; sj[1].reverse = true; sj[1].inclusive = true; sj[1].d_total = slots;
        RC(poly::scan_pair(0, sj, 2, scratch, st));
        HIPCHK(hipMemcpyAsync(h_slots, slots, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        Fr total;
        memcpy(total.d, h_slots, 32);
        RC(poly::mul2c(z, tmp[2], r, n, host::fr_inv(total), st));
        RC(ntt(z, n, BBGPU_IFFT));
        return BBGPU_OK;
    }
    // prover.cpp:88-105
    int compute_z_commitment()
    {
        const uint64_t* sc[1] = { z };
        uint64_t out[1][8];
        PendingCommit P;
        RC(commit_begin(sc, 1, P));
        // beside the commitment: the permutation polynomials need beta and gamma only (prover.cpp:245-247, :253-276)
        RC(copy(sigma[0], sigma_lagrange[0], 3 * n));
        RC(ntt_batch(sigma[0], n, 3, BBGPU_IFFT_WITH_CONSTANT, &challenges.beta));
        for (int k = 0; k < 3; k++) RC(poly::sigma_prepare(s_fft[k], sigma[k], w[k], n, 4 * n, challenges.gamma, st));
        RC(ntt_batch(s_fft[0], 4 * n, 3, BBGPU_COSET_FFT));
        RC(commit_end(P, out));
        memcpy(proof.Z_1, out[0], 64);
        challenges.alpha = challenge(transcript_upto(1)); // compute_alpha, challenge.hpp:87-98
        return BBGPU_OK;
    }
    // prover.cpp:224-300 + :302-341 (fused into one pass over the 4n coset) and :350-402 + arithmetic_widget.cpp:66-104
    int compute_quotient_numerators()
    {
        const size_t n4 = 4 * n, n2 = 2 * n;
        // w_fft (the wires on the 4n coset) and s_fft (w_i + beta sigma_i + gamma there) were enqueued beside the wire / Z commitments
        RC(poly::copy_pad(z_fft, z, n, n4, st)); // :440
        RC(ntt(z_fft, n4, BBGPU_COSET_FFT_WITH_CONSTANT, &challenges.alpha)); // :278
        poly::QuotLargeArgs L{};
        L.wl_f = (const uint32_t*)w_fft[0]; L.wr_f = (const uint32_t*)w_fft[1]; L.wo_f = (const uint32_t*)w_fft[2];
        L.s1_f = (const uint32_t*)s_fft[0]; L.s2_f = (const uint32_t*)s_fft[1]; L.s3_f = (const uint32_t*)s_fft[2];
        L.z_f = (const uint32_t*)z_fft;
        L.q = (uint32_t*)quotient_large;
        L.n4 = (uint32_t)n4;
        RC(poly::quotient_large(L, host::fr_root_of_unity(log2n + 2), challenges.beta, challenges.gamma, st));
        // :446-451: alpha_base = alpha^4 (the product with alpha on :447 is discarded by the reference: fr::mul returns by value)
        const Fr alpha_base = host::fr_sqr(host::fr_sqr(challenges.alpha));
        poly::QuotMidArgs M{};
        M.z_f = (const uint32_t*)z_fft;
        M.wl_f = (const uint32_t*)w_fft[0]; M.wr_f = (const uint32_t*)w_fft[1]; M.wo_f = (const uint32_t*)w_fft[2];
        M.l1 = (const uint32_t*)l_1;
        M.qm_f = (const uint32_t*)q_fft2n[0]; M.ql_f = (const uint32_t*)q_fft2n[1]; M.qr_f = (const uint32_t*)q_fft2n[2];
        M.qo_f = (const uint32_t*)q_fft2n[3]; M.qc_f = (const uint32_t*)q_fft2n[4];
        M.q = (uint32_t*)quotient_mid;
        M.n2 = (uint32_t)n2;
        RC(poly::quotient_mid(M, challenges.alpha, alpha_base, st));
        if (has_seq) { // sequential_widget.cpp:47-62: old_alpha = (alpha_base * alpha) / alpha = the arithmetic widget's own power; hands alpha_base * alpha on unchanged
            poly::QuotSeqArgs Sq{};
            Sq.wo_f = M.wo_f;
            Sq.qon_f = (const uint32_t*)qs_fft2n;
            Sq.q = (uint32_t*)quotient_mid;
            Sq.n2 = (uint32_t)n2;
            RC(poly::quotient_seq(Sq, alpha_base, st));
        }
        if (has_bool) { // the widget chain: the arithmetic widget hands on alpha_base * alpha (arithmetic_widget.cpp:103), the bool widget uses it and the next two powers
            const Fr a5 = host::fr_mul(alpha_base, challenges.alpha), a6 = host::fr_mul(a5, challenges.alpha), a7 = host::fr_mul(a6, challenges.alpha);
            poly::QuotBoolArgs Bq{};
            Bq.wl_f = M.wl_f; Bq.wr_f = M.wr_f; Bq.wo_f = M.wo_f;
            Bq.qbl_f = (const uint32_t*)qb_fft2n[0]; Bq.qbr_f = (const uint32_t*)qb_fft2n[1]; Bq.qbo_f = (const uint32_t*)qb_fft2n[2];
            Bq.q = (uint32_t*)quotient_mid;
            Bq.n2 = (uint32_t)n2;
            RC(poly::quotient_bool(Bq, a5, a6, a7, st));
        }
        if (has_mimc) { // second widget of a MiMCComposer circuit: alpha_base * alpha from the arithmetic widget, alpha_step = alpha
            poly::QuotMimcArgs Mq{};
            Mq.wl_f = L.wl_f; Mq.wr_f = L.wr_f; Mq.wo_f = L.wo_f;
            Mq.qsel_f = (const uint32_t*)qm_fft4n[0]; Mq.qcoef_f = (const uint32_t*)qm_fft4n[1];
            Mq.q = (uint32_t*)quotient_large;
            Mq.n4 = (uint32_t)n4;
            RC(poly::quotient_mimc(Mq, host::fr_mul(alpha_base, challenges.alpha), challenges.alpha, st));
        }
        return BBGPU_OK;
    }
    // prover.cpp:405-465 (after the wire / Z parts above)
    int compute_quotient_polynomial()
    {
        RC(compute_wire_coefficients());
        RC(compute_wire_commitments());
        RC(compute_z_coefficients());
        RC(compute_z_commitment());
        RC(compute_quotient_numerators());
        RC(poly::divide_by_pseudo_vanishing(quotient_mid, log2n, log2n + 1, st));   // :453
        RC(poly::divide_by_pseudo_vanishing(quotient_large, log2n, log2n + 2, st)); // :454
        RC(ntt(quotient_mid, 2 * n, BBGPU_COSET_IFFT));                             // :457
        RC(ntt(quotient_large, 4 * n, BBGPU_COSET_IFFT));                           // :458
        RC(poly::add_inplace(quotient_large, quotient_mid, 2 * n, st));             // :461-463
        return BBGPU_OK;
    }
    // prover.cpp:107-122
    int compute_quotient_commitment()
    {
        const uint64_t* sc[3] = { quotient_large, quotient_large + n * 4, quotient_large + 2 * n * 4 };
        uint64_t out[3][8];
        RC(commit(sc, 3, out));
        memcpy(proof.T_LO, out[0], 64);
        memcpy(proof.T_MID, out[1], 64);
        memcpy(proof.T_HI, out[2], 64);
        challenges.z = challenge(transcript_upto(2)); // compute_evaluation_challenge, challenge.hpp:100-112
        return BBGPU_OK;
    }
    // polynomial_arithmetic.cpp:594-626, l_1 only
    Fr lagrange_l1_at(const Fr& zc) const
    {
        Fr zp = zc;
        for (int i = 0; i < log2n; i++) zp = host::fr_sqr(zp);
        const Fr numerator = host::fr_mul(host::fr_sub(zp, host::fr_one()), host::fr_inv(host::fr_from_u64((uint64_t)n)));
        return host::fr_mul(numerator, host::fr_inv(host::fr_sub(zc, host::fr_one())));
    }
    // prover.cpp:467-538; returns t_eval
    int compute_linearisation_coefficients(Fr* t_eval)
    {
        const Fr& zc = challenges.z;
        const Fr beta_inv = host::fr_inv(challenges.beta);
        const Fr shifted_z = host::fr_mul(zc, host::fr_root_of_unity(log2n));
        // seven evaluations, one read-back (:478-480, :504-506, :512)
        const Fr zs[2] = { zc, shifted_z };
        const poly::EvalJob ej[9] = { { w[0], n, 0, slots + 0 * 4 }, { w[1], n, 0, slots + 1 * 4 }, { w[2], n, 0, slots + 2 * 4 },
                                      { sigma[0], n, 0, slots + 3 * 4 }, { sigma[1], n, 0, slots + 4 * 4 }, { z, n, 1, slots + 5 * 4 },
                                      { quotient_large, 3 * n, 0, slots + 6 * 4 },
                                      // MiMC widget: REQUIRES_W_O_SHIFTED (prover.cpp:499-502) and compute_proof_elements (mimc_widget.cpp:92-95)
                                      { w[2], n, 1, slots + 7 * 4 }, { qm_coeff[1], n, 0, slots + 8 * 4 } };
        const int nev = has_mimc ? 9 : has_seq ? 8 : 7; // the sequential widget is REQUIRES_W_O_SHIFTED too (sequential_widget.cpp:16)
        RC(poly::evaluate_batch_to_device(ej, nev, zs, scratch, st));
        HIPCHK(hipMemcpyAsync(h_slots, slots, (size_t)nev * 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        Fr ev[9];
        memcpy(ev, h_slots, (size_t)nev * 32);
        if (has_mimc || has_seq) proof.w_o_shifted_eval = ev[7];
        if (has_mimc) proof.q_mimc_coefficient_eval = ev[8];
        proof.w_l_eval = ev[0];
        proof.w_r_eval = ev[1];
        proof.w_o_eval = ev[2];
        proof.sigma_1_eval = host::fr_mul(ev[3], beta_inv); // :514-515
        proof.sigma_2_eval = host::fr_mul(ev[4], beta_inv);
        proof.z_1_shifted_eval = ev[5];
        *t_eval = ev[6];

        // linearizer.hpp:29-83
        const Fr k1 = host::fr_from_limbs(FrHostP::GEN5), k2 = host::fr_from_limbs(FrHostP::GEN7);
        const Fr &alpha = challenges.alpha, &beta = challenges.beta, &gamma = challenges.gamma;
        const Fr alpha3 = host::fr_mul(host::fr_sqr(alpha), alpha);
        const Fr zb = host::fr_mul(zc, beta);
        Fr T0 = host::fr_add(host::fr_add(zb, proof.w_l_eval), gamma);
        Fr T1 = host::fr_add(host::fr_add(host::fr_mul(zb, k1), proof.w_r_eval), gamma);
        Fr T2 = host::fr_add(host::fr_add(host::fr_mul(zb, k2), proof.w_o_eval), gamma);
        Fr lt_z1 = host::fr_mul(host::fr_mul(host::fr_mul(T2, T1), T0), alpha);
        T0 = host::fr_add(host::fr_add(host::fr_mul(proof.sigma_1_eval, beta), proof.w_l_eval), gamma);
        T1 = host::fr_add(host::fr_add(host::fr_mul(proof.sigma_2_eval, beta), proof.w_r_eval), gamma);
        Fr lt_sigma3 = host::fr_mul(host::fr_neg(host::fr_mul(host::fr_mul(host::fr_mul(T1, T0), proof.z_1_shifted_eval), alpha)), beta);
        lt_z1 = host::fr_add(lt_z1, host::fr_mul(lagrange_l1_at(zc), alpha3));

        // :520-528 and arithmetic_widget.cpp:106-126 in one pass: r = z_1 Z + (sigma_3 / beta)(beta S_3) + alpha^4 (...selectors...)
        const Fr alpha4 = host::fr_sqr(host::fr_sqr(alpha));
        const Fr w_lr = host::fr_mul(proof.w_l_eval, proof.w_r_eval);
        poly::LinCombArgs A{};
        const uint64_t* ps[10] = { z, sigma[2], q_coeff[0], q_coeff[1], q_coeff[2], q_coeff[3], q_coeff[4], qb_coeff[0], qb_coeff[1], qb_coeff[2] };
        // bool_widget.cpp:106-124: (w^2 - w) alpha^5, alpha^6, alpha^7 on q_bl, q_br, q_bo
        const Fr alpha5 = host::fr_mul(alpha4, alpha), alpha6 = host::fr_mul(alpha5, alpha), alpha7 = host::fr_mul(alpha6, alpha);
        auto boolmul = [](const Fr& e, const Fr& a) { return host::fr_mul(host::fr_sub(host::fr_sqr(e), e), a); };
        const Fr cs[10] = { lt_z1, host::fr_mul(lt_sigma3, beta_inv), host::fr_mul(w_lr, alpha4), host::fr_mul(proof.w_l_eval, alpha4),
                            host::fr_mul(proof.w_r_eval, alpha4), host::fr_mul(proof.w_o_eval, alpha4), alpha4,
                            boolmul(proof.w_l_eval, alpha5), boolmul(proof.w_r_eval, alpha6), boolmul(proof.w_o_eval, alpha7) };
        const uint64_t* pl[12];
        Fr cl[12];
        int terms = 7;
        for (int j = 0; j < 7; j++) { pl[j] = ps[j]; cl[j] = cs[j]; }
        if (has_seq) { // sequential_widget.cpp:64-74: w_o(z omega) * alpha^4 on q_o_next
            pl[terms] = qs_coeff;
            cl[terms++] = host::fr_mul(proof.w_o_shifted_eval, alpha4);
        }
        if (has_bool)
            for (int j = 7; j < 10; j++) { pl[terms] = ps[j]; cl[terms++] = cs[j]; }
        if (has_mimc) { // mimc_widget.cpp:97-113 with alpha_base = alpha^5, alpha_step = alpha
            const Fr t0 = host::fr_add(host::fr_add(proof.w_o_eval, proof.w_l_eval), proof.q_mimc_coefficient_eval);
            const Fr a = host::fr_sub(host::fr_mul(host::fr_sqr(t0), t0), proof.w_r_eval);
            const Fr b = host::fr_mul(host::fr_sub(host::fr_mul(host::fr_sqr(proof.w_r_eval), t0), proof.w_o_shifted_eval), alpha);
            pl[terms] = qm_coeff[0];
            cl[terms++] = host::fr_mul(host::fr_add(b, a), alpha5);
        }
        for (int j = 0; j < terms; j++) A.p[j] = (const uint32_t*)pl[j];
        A.count = terms;
        A.out = (uint32_t*)r;
        A.n = (uint32_t)n;
        RC(poly::lincomb(A, cl, st));
        RC(poly::evaluate(r, n, zc, &proof.linear_eval, scratch, st)); // :536
        return BBGPU_OK;
    }
    // prover.cpp:540-659
    int compute_opening_elements()
    {
        Fr t_eval;
        RC(compute_linearisation_coefficients(&t_eval));
        {
            std::vector<uint64_t> b = transcript_upto(2); // compute_linearisation_challenge, challenge.hpp:114-125
            put_fr(b, proof.w_l_eval); put_fr(b, proof.w_r_eval); put_fr(b, proof.w_o_eval);
            put_fr(b, proof.sigma_1_eval); put_fr(b, proof.sigma_2_eval); put_fr(b, proof.z_1_shifted_eval);
            put_fr(b, proof.linear_eval); put_fr(b, t_eval);
            challenges.nu = challenge(b);
        }
        Fr nu[8];
        nu[0] = challenges.nu;
        for (int i = 1; i < 8; i++) nu[i] = host::fr_mul(nu[i - 1], nu[0]);
        const Fr beta_inv = host::fr_inv(challenges.beta);
        const Fr z_pow_n = host::fr_pow(challenges.z, (uint64_t)n), z_pow_2n = host::fr_pow(challenges.z, (uint64_t)2 * n);
        // :567-595 as one linear combination of nine resident vectors
        poly::LinCombArgs A{};
        // with the MiMC or the sequential widget: w_o joins the shifted opening at nu^8 (prover.cpp:627-635) and q_mimc_coefficient the main one at nu^9
        // (mimc_widget.cpp:115-123)
        const Fr nu9 = host::fr_mul(nu[7], nu[0]);
        const uint64_t* ps[10] = { quotient_large, quotient_large + n * 4, quotient_large + 2 * n * 4, r, w[0], w[1], w[2], sigma[0], sigma[1], qm_coeff[1] };
        const Fr cs[10] = { host::fr_one(), z_pow_n, z_pow_2n, nu[0], nu[1], nu[2], nu[3], host::fr_mul(nu[4], beta_inv), host::fr_mul(nu[5], beta_inv), nu9 };
        const int oterms = has_mimc ? 10 : 9;
        for (int j = 0; j < oterms; j++) A.p[j] = (const uint32_t*)ps[j];
        A.count = oterms;
        A.out = (uint32_t*)tmp[0];
        A.n = (uint32_t)n;
        RC(poly::lincomb(A, cs, st));
        poly::LinCombArgs B{};
        B.p[0] = (const uint32_t*)z;
        B.p[1] = (const uint32_t*)w[2];
        B.count = (has_mimc || has_seq) ? 2 : 1;
        B.out = (uint32_t*)tmp[1];
        B.n = (uint32_t)n;
        RC(poly::lincomb(B, &nu[6], st)); // nu^7 Z (+ nu^8 w_o)
        // compute_kate_opening_coefficients (polynomial_arithmetic.cpp:562-591): W_i = sum_{j>i} F_j z^(j-i-1); the serial
        // recurrence becomes a Horner suffix scan, the remainder F(z) drops out
        const Fr shifted_z = host::fr_mul(challenges.z, host::fr_root_of_unity(log2n));
        poly::ScanJob kj[2] = {};
        kj[0].in = tmp[0]; kj[0].out = tmp[2]; kj[0].n = n; kj[0].reverse = true; kj[0].inclusive = false; kj[0].z = challenges.z;
        kj[1].in = tmp[1]; kj[1].out = r; kj[1].n = n; kj[1].reverse = true; kj[1].inclusive = false; kj[1].z = shifted_z;
        RC(poly::scan_pair(1, kj, 2, scratch, st));
        const uint64_t* sc[2] = { tmp[2], r };
        uint64_t out[2][8];
        RC(commit(sc, 2, out)); // :650-658
        memcpy(proof.PI_Z, out[0], 64);
        memcpy(proof.PI_Z_OMEGA, out[1], 64);
        return BBGPU_OK;
    }
    // waffle::preprocess(prover) (preprocess.hpp:16-55) + ProverArithmeticWidget::compute_preprocessed_commitments
    // (arithmetic_widget.cpp:128-157): the verification key -- commitments to sigma_1..3 and to q_m, q_l, q_r, q_o, q_c
    int preprocess(uint64_t (*out)[8])
    {
        RC(prepare_circuit());
        // sigma polynomials in coefficient form, unscaled (the proof rounds scale them by beta)
        RC(copy(tmp[0], sigma_lagrange[0], n));
        RC(copy(tmp[1], sigma_lagrange[1], n));
        RC(copy(tmp[2], sigma_lagrange[2], n));
        for (int k = 0; k < 3; k++) RC(ntt(tmp[k], n, BBGPU_IFFT));
        const uint64_t* s3[3] = { tmp[0], tmp[1], tmp[2] };
        RC(commit(s3, 3, out));
        const uint64_t* q3[3] = { q_coeff[0], q_coeff[1], q_coeff[2] };
        RC(commit(q3, 3, out + 3));
        const uint64_t* q2[2] = { q_coeff[3], q_coeff[4] };
        RC(commit(q2, 2, out + 6));
        int at = 8;
        if (has_seq) { // sequential_widget.cpp:79-106, second widget of the ExtendedComposer's chain
            const uint64_t* qn[1] = { qs_coeff };
            RC(commit(qn, 1, out + at));
            at += 1;
        }
        if (has_bool) { // bool_widget.cpp:118-152
            const uint64_t* qb[3] = { qb_coeff[0], qb_coeff[1], qb_coeff[2] };
            RC(commit(qb, 3, out + at));
        }
        if (has_mimc) { // mimc_widget.cpp:125-160: q_mimc_coefficient first, then q_mimc_selector
            const uint64_t* qmm[2] = { qm_coeff[1], qm_coeff[0] };
            RC(commit(qmm, 2, out + 8));
        }
        return BBGPU_OK;
    }
    // prover.cpp:661-670
    int construct_proof()
    {
        memset(&proof, 0, sizeof proof);
        timing[0] = timing[1] = timing[2] = 0;
        RC(prepare_circuit());
        const double t0 = now_ms();
        RC(compute_quotient_polynomial());
        RC(compute_quotient_commitment());
        RC(compute_opening_elements());
        timing[0] = now_ms() - t0;
        timing[2] = timing[0] - timing[1];
        return BBGPU_OK;
    }
};

std::mutex g_pmu;
std::vector<PlonkProver*> g_provers;

PlonkProver* get(int h)
{
    if (h < 0 || h >= (int)g_provers.size() || !g_provers[h]) {
        set_error("unknown prover handle %d", h);
        return nullptr;
    }
    return g_provers[h];
}

} // namespace

std::mutex& plonk_mutex() { return g_pmu; }
void plonk_release_all_locked()
{
    for (auto*& p : g_provers) {
        delete p;
        p = nullptr;
    }
    g_provers.clear();
}

} // namespace bbgpu

using namespace bbgpu;

#pragma GCC visibility push(default)
extern "C" {

int bbgpu_plonk_prover_create(const bbgpu_plonk_circuit* c, int srs_handle)
{
    if (!c || !c->w_l || !c->w_r || !c->w_o || !c->sigma_1_mapping || !c->sigma_2_mapping || !c->sigma_3_mapping || !c->q_m || !c->q_l || !c->q_r ||
        !c->q_o || !c->q_c) {
        set_error("null circuit field");
        return BBGPU_ERR_ARG;
    }
    if ((c->q_bl != nullptr) != (c->q_br != nullptr) || (c->q_bl != nullptr) != (c->q_bo != nullptr)) {
        set_error("bool widget selectors: give all of q_bl, q_br, q_bo or none");
        return BBGPU_ERR_ARG;
    }
    if ((c->q_mimc_selector != nullptr) != (c->q_mimc_coefficient != nullptr) || (c->q_mimc_selector && (c->q_bl || c->q_o_next))) {
        set_error("MiMC widget selectors: give both q_mimc_selector and q_mimc_coefficient or neither, and not together with the bool or the sequential widget");
        return BBGPU_ERR_ARG;
    }
    // 2^21: the largest circuit with a REFERENCE proof to compare against (tests/golden/plonk_proofs.json); nothing in the kernels stops there any more -- the
    // scans nest to 2^28 elements (poly.hip, round 4), the transforms reach 2^28, the commitments run over table segments -- but a proof of a larger circuit
    // would be parity-unpinned, so the entry refuses it instead of returning bytes no reference ever produced.
    if (c->n < 4 || (c->n & (c->n - 1)) || c->n > ((size_t)1 << 21)) {
        set_error("circuit size %zu: must be a power of two, 4 <= n <= 2^21 (the largest size a reference proof exists for: larger proofs would be parity-unpinned)", c->n);
        return BBGPU_ERR_SIZE;
    }
    const int W = bbgpu_srs_num_windows(srs_handle, c->n);
    if (W < 0) {
        set_error("unknown SRS handle %d", srs_handle);
        return BBGPU_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_pmu);
    if (int rcb = bind_calling_thread()) return rcb; // the kernels below are launched from THIS thread
    PlonkProver* p = new PlonkProver();
    int rc = p->init(c, srs_handle);
    if (rc) {
        delete p;
        return rc;
    }
    g_provers.push_back(p);
    return (int)g_provers.size() - 1;
}

int bbgpu_plonk_prover_set_witness(int prover, const uint64_t* w_l, const uint64_t* w_r, const uint64_t* w_o)
{
    std::lock_guard<std::mutex> lk(g_pmu);
    if (int rcb = bind_calling_thread()) return rcb; // the kernels below are launched from THIS thread
    PlonkProver* p = get(prover);
    if (!p || !w_l || !w_r || !w_o) return BBGPU_ERR_ARG;
    return p->set_witness(w_l, w_r, w_o);
}

int bbgpu_plonk_construct_proof(int prover, uint64_t proof_out[BBGPU_PLONK_PROOF_WORDS])
{
    std::lock_guard<std::mutex> lk(g_pmu);
    if (int rcb = bind_calling_thread()) return rcb; // the kernels below are launched from THIS thread
    PlonkProver* p = get(prover);
    if (!p || !proof_out) return BBGPU_ERR_ARG;
    int rc = p->construct_proof();
    if (rc) return rc;
    memcpy(proof_out, &p->proof, sizeof(Proof));
    return BBGPU_OK;
}

int bbgpu_plonk_preprocess(int prover, uint64_t vk_out[BBGPU_PLONK_VK_WORDS])
{
    std::lock_guard<std::mutex> lk(g_pmu);
    if (int rcb = bind_calling_thread()) return rcb; // the kernels below are launched from THIS thread
    PlonkProver* p = get(prover);
    if (!p || !vk_out) return BBGPU_ERR_ARG;
    return p->preprocess(reinterpret_cast<uint64_t(*)[8]>(vk_out));
}

int bbgpu_plonk_last_challenges(int prover, uint64_t out[20])
{
    std::lock_guard<std::mutex> lk(g_pmu);
    PlonkProver* p = get(prover);
    if (!p || !out) return BBGPU_ERR_ARG;
    memcpy(out, &p->challenges, 160);
    return BBGPU_OK;
}

int bbgpu_plonk_last_timing(int prover, double ms_out[4])
{
    std::lock_guard<std::mutex> lk(g_pmu);
    PlonkProver* p = get(prover);
    if (!p || !ms_out) return BBGPU_ERR_ARG;
    for (int i = 0; i < 4; i++) ms_out[i] = p->timing[i];
    return BBGPU_OK;
}

// challenge.hpp:64-112 on a finished proof: gamma, beta, alpha and the evaluation challenge z (what a verifier recomputes).
// Host arithmetic only; needs no GPU.
int bbgpu_plonk_challenges_from_proof(const uint64_t proof_words[BBGPU_PLONK_PROOF_WORDS], uint64_t out[16])
{
    if (!proof_words || !out) return BBGPU_ERR_ARG;
    PlonkProver p;
    memcpy(&p.proof, proof_words, sizeof(Proof));
    std::vector<uint64_t> b = p.transcript_upto(0);
    const Fr gamma = PlonkProver::challenge(b);
    PlonkProver::put_fr(b, gamma);
    const Fr beta = PlonkProver::challenge(b);
    const Fr alpha = PlonkProver::challenge(p.transcript_upto(1));
    const Fr z = PlonkProver::challenge(p.transcript_upto(2));
    memcpy(out, gamma.d, 32);
    memcpy(out + 4, beta.d, 32);
    memcpy(out + 8, alpha.d, 32);
    memcpy(out + 12, z.d, 32);
    return BBGPU_OK;
}

int bbgpu_plonk_prover_destroy(int prover)
{
    std::lock_guard<std::mutex> lk(g_pmu);
    PlonkProver* p = get(prover);
    if (!p) return BBGPU_ERR_ARG;
    delete p;
    g_provers[prover] = nullptr;
    return BBGPU_OK;
}

} // extern "C"
#pragma GCC visibility pop
