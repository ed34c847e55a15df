// poly.h -- internal interface of poly.hip (device-resident Fr polynomial helpers), used by plonk.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fe.hpp"
#include "host_fr.hpp"

namespace bbgpu {
namespace poly {

struct PowTab {
    Limbs9 p[24]; // base^(2^j), 2^261 form
};

// growable device workspace private to one stream of work (+ 4 KiB of pinned host memory for 32-byte read-backs)
struct Scratch {
    uint8_t* base = nullptr;
    size_t cap = 0;
    void* h_pinned = nullptr;
    int ensure(size_t bytes);
    void release();
};

struct ZTermsArgs {
    const uint32_t *w_l, *w_r, *w_o, *s1, *s2, *s3; // n Lagrange-form values each
    uint32_t *num, *den;
    uint32_t n;
    PowTab root;
    Limbs9 step_m261, beta_m256, beta_m261, beta_k1_m256, beta_k2_m256, gamma_m256;
};
struct QuotLargeArgs {
    const uint32_t *wl_f, *wr_f, *wo_f, *s1_f, *s2_f, *s3_f, *z_f; // 4n coset evaluations each
    uint32_t* q;
    uint32_t n4;
    PowTab root;
    Limbs9 g_m261, step_m261, beta_m256, beta_k1_m256, beta_k2_m256, gamma_m256;
};
struct QuotMidArgs {
    const uint32_t *z_f, *wl_f, *wr_f, *wo_f;           // 4n coset evaluations
    const uint32_t *l1;                                  // 2n
    const uint32_t *qm_f, *ql_f, *qr_f, *qo_f, *qc_f;    // 2n coset evaluations of the selectors, unscaled
    uint32_t* q;
    uint32_t n2;
    Limbs9 alpha_m256, alpha_fix_m261, alpha2_fix_m261, abase_m261, abase_fix2_m261, abase_fix3_m261;
};
struct QuotSeqArgs {
    const uint32_t* wo_f;   // 4n coset evaluations of w_o (read at index 2i + 4: the next gate's output wire)
    const uint32_t* qon_f;  // 2n coset evaluations of q_o_next, unscaled
    uint32_t* q;            // quotient_mid, accumulated into
    uint32_t n2;
    Limbs9 c_fix_m261;
};
struct QuotBoolArgs {
    const uint32_t *wl_f, *wr_f, *wo_f;    // 4n coset evaluations (read at index 2i)
    const uint32_t *qbl_f, *qbr_f, *qbo_f; // 2n coset evaluations of the bool selectors, unscaled
    uint32_t* q;                           // quotient_mid, accumulated into
    uint32_t n2;
    Limbs9 cl_fix_m261, cr_fix_m261, co_fix_m261; // alpha powers times 2^5
};
struct QuotMimcArgs {
    const uint32_t *wl_f, *wr_f, *wo_f; // 4n coset evaluations
    const uint32_t *qsel_f, *qcoef_f;   // 4n coset evaluations of q_mimc_selector / q_mimc_coefficient, unscaled
    uint32_t* q;                        // quotient_large, accumulated into
    uint32_t n4;
    Limbs9 alpha_m261, abase_fix_m261;  // alpha_step; alpha_base * 2^5
};
struct LinCombArgs {
    const uint32_t* p[12];
    Limbs9 c[12];
    const uint32_t* out_add; // optional addend (memory form), may be null
    uint32_t* out;
    uint32_t n;
    int count;
};

struct EvalJob {
    const uint64_t* coeffs;
    size_t n;
    int zsel;           // evaluate at z[zsel]
    uint64_t* d_result; // 32-byte device slot
};
struct EvalBatchArgs {
    const uint32_t* c[10];
    uint32_t* result[10];
    uint32_t n[10], blocks[10];
    uint8_t zsel[10];
    Limbs9 zT[10];
    PowTab T[2];
    uint32_t* partial; // 10 x 256 elements
};
struct ScanJob {
    const uint64_t* in;
    uint64_t* out;      // may be null when only the total is wanted
    size_t n;
    bool reverse, inclusive;
    host::Fr z;         // Horner scans only
    uint64_t* d_total;  // optional 32-byte device slot
};

PowTab make_powtab(const host::Fr& base);
// up to two scans of the same kind in shared launches: mode 0 = running products, mode 1 = Horner suffix sums
int scan_pair(int mode, const ScanJob* jobs, int count, Scratch& S, hipStream_t st);
int evaluate_batch_to_device(const EvalJob* jobs, int count, const host::Fr z[2], Scratch& S, hipStream_t st);

int powers(uint64_t* d_out, size_t n, const host::Fr& base, const host::Fr& start, hipStream_t st);
int copy_pad(uint64_t* d_dst, const uint64_t* d_src, size_t n_src, size_t n_dst, hipStream_t st);
int add_inplace(uint64_t* d_a, const uint64_t* d_b, size_t n, hipStream_t st);
int mul_pointwise(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, hipStream_t st);
int mul2c(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, const host::Fr& c, hipStream_t st);

size_t scan_scratch_bytes(size_t n);
// exclusive / inclusive running products, prefix or suffix; d_out may be null when only the total is wanted
int product_scan(const uint64_t* d_in, uint64_t* d_out, size_t n, bool reverse, bool inclusive, Scratch& S, hipStream_t st, uint64_t* d_total);
// out_i = sum_{j >= i (inclusive) or j > i} in_j z^(j - i [- 1])
int horner_suffix(const uint64_t* d_in, uint64_t* d_out, size_t n, const host::Fr& z, bool inclusive, Scratch& S, hipStream_t st, uint64_t* d_total);
int evaluate_to_device(const uint64_t* d_coeffs, size_t n, const host::Fr& z, uint64_t* d_result, Scratch& S, hipStream_t st);
int evaluate(const uint64_t* d_coeffs, size_t n, const host::Fr& z, host::Fr* out, Scratch& S, hipStream_t st);
int batch_invert(uint64_t* d_v, uint64_t* d_tmp, size_t n, Scratch& S, hipStream_t st);

int sigma_from_mapping(uint64_t* d_out, const uint32_t* d_mapping, const uint64_t* d_roots, size_t n, hipStream_t st);
int z_terms(ZTermsArgs A, const host::Fr& root, const host::Fr& beta, const host::Fr& gamma, hipStream_t st);
int sigma_prepare(uint64_t* d_dst, const uint64_t* d_sigma, const uint64_t* d_w, size_t n, size_t n_dst, const host::Fr& gamma, hipStream_t st);
int quotient_large(QuotLargeArgs A, const host::Fr& root4n, const host::Fr& beta, const host::Fr& gamma, hipStream_t st);
int quotient_mid(QuotMidArgs A, const host::Fr& alpha, const host::Fr& alpha_base, hipStream_t st);
int quotient_mimc(QuotMimcArgs A, const host::Fr& alpha_base, const host::Fr& alpha_step, hipStream_t st);
int quotient_bool(QuotBoolArgs A, const host::Fr& c_left, const host::Fr& c_right, const host::Fr& c_out, hipStream_t st);
int quotient_seq(QuotSeqArgs A, const host::Fr& c, hipStream_t st); // sequential_widget.cpp:47-62: quotient_mid[i] += c q_o_next[i] w_o[2i + 4]
int divide_by_pseudo_vanishing(uint64_t* d_coeffs, int log2n, int log2N, hipStream_t st);
int lagrange_l1_fft(uint64_t* d_l1, uint64_t* d_tmp, int log2n, int log2N, Scratch& S, hipStream_t st);
int lincomb(LinCombArgs A, const host::Fr* coeffs, hipStream_t st);

} // namespace poly
} // namespace bbgpu
