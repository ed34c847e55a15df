/*
 * bbgpu.h -- C ABI of libbbgpu.so: MI355X (gfx950) implementations of barretenberg's PLONK-prover hot path,
 * BN254 G1 Pippenger MSM and radix-2 NTT / coset-FFT over Fr.
 *
 * Every entry point names the reference interface it replaces (paths under /root/reference/src/barretenberg/).
 * Conventions are the reference's own:
 *   field element   = 4 x uint64_t little-endian limbs, Montgomery form (x * 2^256 mod p)      fields/field.hpp:19-22
 *   affine G1 point = {x, y} = 8 x uint64_t; Jacobian = {x, y, z} = 12 x uint64_t               groups/group.hpp:17-28
 *   point at infinity <=> bit 63 of y limb 3                                                    groups/group.hpp:133-151
 *   scalars may be any representative in [0, 2r); NTT inputs in [0, 2^256); NTT outputs canonical [0, r)
 *   MSM results are returned NORMALISED: z = fq::one, x,y canonical (what batched_scalar_multiplications hands the
 *   prover, scalar_multiplication.cpp:765; any Jacobian representative is legal for pippenger(), :457-476)
 * All functions return BBGPU_OK (0) or a negative error code; bbgpu_last_error() describes the last failure of the
 * calling thread.  The GPU entry points have NO CPU fallback: if no GPU / no code object / no memory, they fail loudly.  The reference's
 * C++ signatures cannot report an error, so the C++ shim -- and only it -- answers a failed call with the bbgpu_host_* entries at the
 * end of this header (SURVEY 8b "the C++ shim must turn non-zero into CPU fallback"); BBGPU_SHIM_STRICT=1 makes it abort instead.
 */
#ifndef BBGPU_H
#define BBGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    BBGPU_OK = 0,
    BBGPU_ERR_HIP = -1,   /* a HIP runtime call failed (no device, out of memory, launch failure) */
    BBGPU_ERR_SIZE = -2,  /* size not supported (NTT: n must be 2^k, 2 <= n <= 2^28 = the two-adicity of the field) */
    BBGPU_ERR_ARG = -3,   /* null pointer / bad enum / unknown handle */
    BBGPU_ERR_STATE = -4, /* library not initialised / every MSM slot in flight / a table that lacks what the call needs */
    BBGPU_ERR_LOST = -5   /* an IN-PLACE host-buffer call failed while its result was being copied back: the caller's buffer may hold a mixture of
                             input and output (the one failure the shim cannot answer with a host computation) */
};

/* polynomial_arithmetic.hpp:27-41: which member of the fft family */
typedef enum {
    BBGPU_FFT = 0,                     /* fft()                      polynomial_arithmetic.cpp:266 */
    BBGPU_IFFT = 1,                    /* ifft()                     :271-277 */
    BBGPU_COSET_FFT = 2,               /* coset_fft()                :287-291 */
    BBGPU_COSET_IFFT = 3,              /* coset_ifft()               :311-315 */
    BBGPU_FFT_WITH_CONSTANT = 4,       /* fft_with_constant()        :279-285 */
    BBGPU_IFFT_WITH_CONSTANT = 5,      /* ifft_with_constant()       :301-309 */
    BBGPU_COSET_FFT_WITH_CONSTANT = 6  /* coset_fft_with_constant()  :293-299 */
} bbgpu_ntt_kind;

/* ---- lifetime ---------------------------------------------------------------------------------------------------- */
int bbgpu_init(int device);       /* binds the calling process to one GPU (one process per GPU), allocates workspaces */
void bbgpu_shutdown(void);
int bbgpu_device_count(void);
const char* bbgpu_last_error(void);
const char* bbgpu_version(void);

/* Device memory the library holds for the life of the process ("one-time" state: excluded from every timed figure, but a shared GPU has to plan for it).
 * Everything here is bounded: the point tables registered on first sight by BBGPU_SRS_CACHE_BYTES (16 GiB, least recently used first; explicitly registered
 * ones until bbgpu_srs_release), the window tables of one SRS by BBGPU_TABLE_MAX_BYTES (64 GiB), the transforms' twiddle / twist tables -- 128 MiB per 2^20
 * domain, 512 MiB per 2^22 -- by BBGPU_NTT_TABLE_BYTES (8 GiB, least recently used domain sizes dropped and rebuilt on demand), the workspaces by the largest
 * call seen (eight MSM slots, scalar staging, transform scratch). */
typedef struct {
    uint64_t srs_points_bytes;    /* resident base points of all live tables */
    uint64_t srs_table_bytes;     /* their pre-shifted window tables */
    uint64_t srs_auto_bytes;      /* of the two above: held by tables registered on first sight (evictable under srs_cache_cap_bytes) */
    uint64_t srs_cache_cap_bytes;
    uint64_t ntt_table_bytes;     /* twiddle / twist / coset tables of the cached domain sizes */
    uint64_t ntt_table_cap_bytes;
    uint64_t ntt_table_sets;      /* how many domain sizes are cached */
    uint64_t msm_workspace_bytes; /* the MSM slots' workspaces (allocated on first use of a slot, sized by its largest MSM) */
    uint64_t staging_bytes;       /* scalar / coefficient staging, transform scratch, polynomial temporaries */
    uint64_t pinned_host_bytes;   /* pinned host memory: staging buffers and the slots' result arrays */
} bbgpu_memory_info;
int bbgpu_memory_stats(bbgpu_memory_info* out);

/* ---- the error contract on a machine that HAS a GPU: fault injection (testing) ---------------------------------------
 * The reference API has no error channel (assert.hpp:19-23 compiles to nothing, scalar_multiplication.cpp:680-684 prints and returns), so a GPU call
 * that fails in the middle of a proof must leave the library usable and the shim able to answer on the host (SURVEY 8b "Errors").  Every device
 * allocation, every copy of a caller's buffer and every launch check of the library passes one of four host-side funnels; a spec
 *     "alloc:k" | "h2d:k" | "d2h:k" | "launch:k"          (k = 0: the next such call)
 * makes the k-th call of that kind -- counted from the moment the spec is set -- fail ONCE with the error a real failure of that kind returns
 * (out of memory / invalid value / launch failure) without touching the device.  The environment variable BBGPU_FAIL_AT holds the same spec for
 * programs that cannot call this (read once, at the library's first allocation / copy / launch).  NULL or "" disarms.  No kernel reads any of it.
 * bbgpu_fault_stats: how often each funnel was passed since the spec was set, whether the armed failure fired, how many failures the library rode
 * out by itself (an SRS kept without its window tables), the library's live device allocations (count and bytes: after bbgpu_shutdown() both
 * are 0 unless an error path leaked) and the MSM slots still in flight. */
typedef struct {
    uint64_t alloc_calls, h2d_calls, d2h_calls, launch_checks;
    uint64_t armed;            /* 1 while a failure is armed and has not fired yet */
    uint64_t fired;            /* failures injected since the spec was set (0 or 1) */
    uint64_t absorbed;         /* failures the library continued after WITHOUT returning an error (degraded, still on the GPU) */
    uint64_t live_allocations; /* device allocations of the library that are live now */
    uint64_t live_bytes;
    uint64_t slots_pending;    /* MSM slots with work in flight or a result not yet collected */
} bbgpu_fault_info;
int bbgpu_fault_inject(const char* spec);
int bbgpu_fault_stats(bbgpu_fault_info* out);

/* ---- NTT ---------------------------------------------------------------------------------------------------------
 * Drop-in for polynomial_arithmetic::{fft,ifft,coset_fft,coset_ifft,fft_with_constant,ifft_with_constant,
 * coset_fft_with_constant}(fr::field_t* coeffs, const evaluation_domain& domain[, const fr::field_t& constant]):
 * transforms coeffs[0..n) in place.  `constant` (4 limbs, Montgomery) is read for the *_with_constant kinds only.
 * bbgpu_ntt: host buffer (copied to the device and back; n <= 16 is answered on the host, see bbgpu_set_host_thresholds).
 * bbgpu_ntt_device: device-resident buffer, asynchronous on `hip_stream` (a hipStream_t; NULL = the legacy default stream, so a
 * caller working on the null stream is ordered with the transform as with its own kernels).  The same holds
 * for the polynomial helpers below.  The MSM device entries differ: there NULL selects the ticket's own internal NON-BLOCKING stream
 * (that is what lets consecutive MSMs overlap), which has no implicit ordering with the caller's null stream -- d_scalars must be
 * complete before the call (synchronise, or pass the producing stream).
 * Calls on different streams may be in flight together: the library's shared scratch is handed from one stream to the next by an
 * event (they serialise on the device, results are independent of the interleaving). */
int bbgpu_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant);
int bbgpu_ntt_device(uint64_t* d_coeffs, size_t n, int kind, const uint64_t* constant, void* hip_stream);
/* `batch` (<= 64) transforms of the same size and kind in one set of launches; transform j occupies
 * d_coeffs[j * stride_elems .. j * stride_elems + n) (the prover transforms its three wire / sigma polynomials together) */
int bbgpu_ntt_device_batch(uint64_t* d_coeffs, size_t n, size_t stride_elems, int batch, int kind, const uint64_t* constant, void* hip_stream);

/* ---- MSM ---------------------------------------------------------------------------------------------------------
 * The prover passes the same SRS to every MSM (reference_string.cpp:16-35), laid out as the 2n-entry endomorphism
 * table of generate_pippenger_point_table (scalar_multiplication.cpp:131-140): entry 2i = P_i, entry 2i+1 = (beta x_i,
 * -y_i).  bbgpu_srs_register uploads the n base points (even entries) once and keeps them resident on the GPU in the
 * kernels' working form; it returns a handle >= 0.  Host-pointer MSM calls look the table up by address (and register
 * it on first sight), so sub-slices `points + 2*off` of a registered table are served from the resident copy
 * (batched_scalar_multiplications slices exactly like that, scalar_multiplication.cpp:720-726). */
/* The address is only a hint: every address hit is re-validated against a per-point content fingerprint taken at upload (first, last
 * and 14 evenly spaced rows of the range the caller passes, the spaced rows moving on with every check; only memory inside that range
 * is read).  RESIDUAL WINDOW: a table rewritten IN PLACE only partially -- first and last row unchanged -- is served from the stale
 * resident copy until a sampled row falls into the rewritten part (after k checks a rewritten fraction f survives with probability
 * ~(1 - f)^(14 k)); a caller that edits a slice of a live SRS must release / re-register it.  A table that was registered on
 * first sight and whose memory now holds other points (freed and reused, or refilled in place) is evicted and uploaded again;
 * tables registered on first sight are also evicted least-recently-used beyond BBGPU_SRS_CACHE_BYTES (default 16 GiB of device
 * memory).  A table registered EXPLICITLY keeps its handle until bbgpu_srs_release: mutate it in place only after releasing it
 * (host-pointer calls stop being served from a handle whose contents changed, device-pointer calls by handle cannot notice). */
int bbgpu_srs_register(const uint64_t* points_endo_table, size_t n);
/* EXACT mode for the address-keyed cache (the reference reads the caller's points on every call, scalar_multiplication.cpp:604-617): with
 * full != 0 every host-pointer MSM (bbgpu_msm_g1, bbgpu_msm_g1_batch and the shim entries above them) served from this table re-hashes EVERY row of
 * the range it uses against the fingerprints taken at upload -- on the host, spread over the staging threads, while the call's kernels already run
 * against the resident copy; if a single row differs the copy is dropped (a table registered on first sight is evicted, an explicitly registered one
 * stops serving host-pointer calls; its handle stays valid) and the call runs once more on a fresh upload: identical inputs -> identical outputs on
 * the very next call, whatever part of the table was rewritten.  srs_handle -1 sets the default for tables registered from now on (on first sight
 * or explicitly); the environment variable BBGPU_SRS_VALIDATE=full does the same at start-up.  A table large enough that the call's scalars bypass the staging
 * pool (more than 8 MiB of them: 2^18 points) is checked in the BACKGROUND on the pool's helper threads, from before the upload of the scalars to after the launches.
 * Cost on MI355X + EPYC 9575F (tools/validate_ab.py, profiles/r05_boundary_ab.txt, one box each, alternating): bbgpu_msm_g1 at 2^16 points +0.1 % (0.35 ms either
 * way); at 2^20 points -- 64 MiB of host memory hashed per call -- +5 ... 11 % by box with the default three helper threads (1.70 -> 1.78 ... 1.87 ms; +21 % before
 * the check moved to the background), +2.7 % with BBGPU_STAGE_THREADS=7, +0.1 % with 11; a batch of three 2^20-point jobs over one table +0.8 % (one check per
 * distinct range).  At or above 5 % at the headline size with the default threads, so the DEFAULT STAYS THE 16-ROW SAMPLE and the residual window described above
 * stays with it; a caller that rewrites live tables in place sets the flag (and gives the pool more threads if it has the cores). */
int bbgpu_srs_set_validate(int srs_handle, int full);
/* Registration also builds, on the device, the pre-shifted window tables 2^(c w) * P_i (the reference's
 * generate_pippenger_precompute_table idea, scalar_multiplication.cpp:90-129): W x n x 64 bytes (1 GiB at n = 2^20), so that
 * all digit windows share one bucket set.  On by default from 1024 points on; bbgpu_set_precompute(0) turns it off for
 * tables registered afterwards.  Results are identical either way.
 * The sorted entries of an MSM carry a 24-bit table row, i.e. one table serves 2^24 / W points (2^20 at 15 windows): a LARGER SRS keeps
 * one table per segment of at most that many points (equal segments; 16 GiB of tables at 2^24 points, up to BBGPU_TABLE_MAX_BYTES = 64 GiB,
 * beyond which -- or when the allocation fails -- the points stay resident without tables), and an MSM over it runs as one PIECE per
 * segment it touches, dealt to the ticket's slot and a helper slot, the piece sums added on the host: the reference's own decomposition
 * into point ranges per thread (scalar_multiplication.cpp:703-738).  2^21 / 2^22 / 2^24 points: 2.7 / 5.1 / 18.9 ms on one MI355X.
 * Row- and bucket-range shares (below) are defined on ONE segment. */
void bbgpu_set_precompute(int enabled);
/* Multi-GPU: tables registered after this call keep only the digit windows that rank `rank` of `world` touches when the W x n (window,
 * point) rows are split evenly over the ranks (bbgpu_msm_g1_device_rows_async with rows [W n r / N, W n (r + 1) / N), or whole-window
 * shares inside that range): ceil(W / world) + 1 windows instead of W.  Asking such a table for other windows returns BBGPU_ERR_STATE.
 * (0, 1) restores full tables. */
void bbgpu_set_table_share(int rank, int world);
/* Multi-GPU, split by POINT range (the slicing of scalar_multiplication.cpp:703-738 -- ranges of points per thread, summed at the end -- at the
 * multi-GPU level): rank r of N registers points [n r / N, n (r + 1) / N) as its own SRS and runs bbgpu_msm_g1_device(_async) over the matching
 * scalars; the N results add up to the MSM (bbgpu_g1_sum).  After this call, tables registered pick the window size the WHOLE MSM of
 * world * n points would (17 bits from 2^19 points on) instead of the one for n points.  1 restores the default. */
void bbgpu_set_point_share(int world);
/* number of digit windows an MSM of n points against this table is split into (use this, not bbgpu_msm_num_windows, to
 * shard windows over ranks: a table carries the window size it was built for) */
int bbgpu_srs_num_windows(int srs_handle, size_t n);
int bbgpu_srs_release(int handle);
/* resident tables right now: all, those registered on first sight (evictable), and the device bytes the latter hold (any may be NULL) */
int bbgpu_srs_cache_stats(int* live_entries, int* auto_entries, uint64_t* auto_bytes);
/* io::read_transcript, G1 part (io/io.hpp:36-182): reads `degree - 1` points of an ignition-format transcript file
 * (srs_db/transcript.dat) behind the generator and writes the 2 * degree entry endomorphism table of
 * generate_pippenger_point_table -- the `monomials` array of ReferenceString (reference_string.cpp:16-35) -- ready for
 * bbgpu_srs_register.  Host code; needs no GPU. */
int bbgpu_transcript_read_g1(const char* path, size_t degree, uint64_t* points_endo_table_out);
/* The writer of the same format: the file io::read_transcript(monomials, g2_x, degree, path) accepts for the SRS whose endo table is
 * given -- degree - 1 G1 points (the generator is implicit), then G2 and x * G2 (the verifier's pairing input, io.hpp:171-180) computed
 * here from the secret `x_mont` (Montgomery), then the 64-byte checksum slot the reference never verifies.  Together with
 * bbgpu_srs_generate this stands in for the missing srs_db/transcript.dat of BASELINE configs 2 and 5.  Host code; needs no GPU. */
int bbgpu_transcript_write(const char* path, const uint64_t* points_endo_table, size_t degree, const uint64_t x_mont[4]);
/* device-side generation of the synthetic SRS x^i * G, i < n, straight into a resident table; optionally also written
 * back to the host as the reference-format 2n endo table (may be NULL).  Stands in for the missing srs_db/transcript.dat */
int bbgpu_srs_generate(const uint64_t* x_mont, size_t n, uint64_t* host_endo_table_out);
/* the same for the points x^(first + i) * G, i < n: the slice a rank of an N-way POINT-range split of a larger MSM keeps (rank r of N:
 * first = r n / N; its MSM over the matching slice of the scalars is its partial sum, bench.py --shard points) */
int bbgpu_srs_generate_range(const uint64_t* x_mont, size_t first, size_t n, uint64_t* host_endo_table_out);

/* drop-in for scalar_multiplication::pippenger(scalars, points, n, bucket_width) (:457-476); scalars not modified.
 * out = {x, y, z} normalised, or infinity flag set (n == 0, all-zero scalars).
 * A sum that IS the point at infinity comes back as the clean encoding (all limbs zero, bit 63 of y limb 3 set).  What the reference emits
 * for it is pinned by tests/golden/infinity_commitments.json: g1::normalize() re-sets the flag (group.hpp:450-468) and every other bit of the
 * pair is whatever the CPU algorithm's accumulators held in that run -- it changes with the OpenMP thread count, and through the Fiat-Shamir
 * hash so does the rest of such a proof.  The flag is all there is to reproduce; the reference's Verifier accepts proofs carrying the clean
 * encoding (tests/test_gpu_plonk.py::test_commitments_at_infinity).  Every other result is the unique affine point and is bit-identical.
 * From 2^19 points on the call runs as two point ranges through the two-slot pipeline (the second range's scalars cross the link under the
 * first one's kernels; BBGPU_HOST_MSM_SPLIT).  Host buffers of up to 8 MiB (BBGPU_STAGE_MAX_BYTES) are copied through the library's own
 * pinned staging buffers rather than pinned in place by the runtime (DESIGN_HISTORY.md 1). */
int bbgpu_msm_g1(const uint64_t* scalars, const uint64_t* points_endo_table, size_t n, uint64_t out[12]);
/* the same sum over a PLAIN table: `points` = n affine points, 64 bytes apart -- the argument convention of the reference's
 * pippenger_low_memory(scalars, points, num_points) (scalar_multiplication.cpp:142-262, which applies beta itself; its test allocates
 * exactly n * 64 bytes, test_scalar_multiplication.cpp:164-187) and of round_points.back() in pippenger_precomputed (:478-574,
 * test_scalar_multiplication.cpp:226-262).  Reads exactly n * 64 bytes of `points`; the table is used once and not cached (these
 * are the reference's test / bench entries, not the prover's).  scalars not modified. */
int bbgpu_msm_g1_plain(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t out[12]);
/* SURVEY 8b "small sizes": the reference's callers include the Verifier's per-proof MSM over ~20 freshly built points
 * (verifier.cpp:359-363) and proofs of n = 4 circuits (test_verifier.cpp:105-122).  Host-pointer MSMs of at most `msm_max_points`
 * points against a table that is not resident, and host-buffer transforms (bbgpu_ntt) of at most `ntt_max_elements` (<= 64)
 * elements, are answered on the host by csrc/host_small.hpp -- no device allocation, copy or launch; results identical.  Defaults
 * 24 / 16 (env BBGPU_HOST_MSM_MAX / BBGPU_HOST_NTT_MAX); 0 / 0 sends every size to the GPU.  Nothing larger ever runs on the host. */
void bbgpu_set_host_thresholds(int msm_max_points, int ntt_max_elements);

/* drop-in for scalar_multiplication::batched_scalar_multiplications(mul_state, num) (:650-772); layout-identical to
 * multiplication_state (scalar_multiplication.hpp:88-94: points@0, scalars@8, num_elements@16, output@32, size 128) */
typedef struct {
    const uint64_t* points;  /* 2n-entry endo table */
    const uint64_t* scalars; /* n scalars */
    size_t num_elements;
    uint64_t _pad;
    uint64_t output[12]; /* written normalised */
} bbgpu_msm_job;
int bbgpu_msm_g1_batch(bbgpu_msm_job* jobs, size_t num_jobs);

/* scalars already resident in HBM (n x 4 limbs), points = a registered SRS handle (first n points, starting at
 * point `offset`).  Windows [window_begin, window_end) of the signed-digit decomposition are processed -- the whole
 * scalar is windows [0, bbgpu_msm_num_windows(n)).  The result is the partial sum over those windows, normalised; partial
 * sums of disjoint window ranges add up to the full MSM (bbgpu_g1_sum).  This is the multi-GPU entry: rank g takes its
 * share of the windows, partial sums are exchanged (RCCL all-gather of 96 bytes per rank) and folded on every rank. */
int bbgpu_msm_num_windows(size_t n);
int bbgpu_msm_g1_device(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int window_begin,
                        int window_end, uint64_t out[12], void* hip_stream);
/* Asynchronous form: enqueue (returns a ticket >= 0, or a negative error) and collect later.  A ticket is waited for once, by one
 * thread; the wait blocks outside the library's mutex, so other threads keep issuing and collecting meanwhile.  Up to eight MSMs may be in
 * flight; the bucket-reduction tail and host finish of one then overlap the sort/accumulate of the next (DESIGN_HISTORY.md 5), and
 * small latency-bound MSMs (a prover round's three commitments) run side by side.  With hip_stream == NULL each ticket runs
 * on its own internal stream.  An MSM enqueued while another is in flight is laid out for throughput instead of latency (longer
 * accumulation chunks, row / column sums in two steps: DESIGN_HISTORY.md 5, 6 v); the result is the same point either way. */
int bbgpu_msm_g1_device_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int window_begin,
                              int window_end, void* hip_stream);
/* The same with a share that may start and end INSIDE a digit window: rows [row_begin, row_end) of the W x n (window, point) pairs counted
 * window-major, row = w * n + i, 0 <= row_begin < row_end <= W * n with W = bbgpu_srs_num_windows().  Against window tables every pair is
 * one table row feeding the one shared bucket set, so any split of the rows splits the MSM; N ranks taking [W n r / N, W n (r + 1) / N)
 * stay balanced when N does not divide W.  Needs the window tables (bbgpu_srs_has_window_tables() == 1), else BBGPU_ERR_STATE.
 * Collected with bbgpu_msm_g1_wait like any other ticket; the partial sums of a complete split add up to the MSM (bbgpu_g1_sum). */
int bbgpu_msm_g1_device_rows_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, uint64_t row_begin, uint64_t row_end,
                                   void* hip_stream);
/* The other way to split one MSM over N ranks (against window tables): share s of N keeps the digits whose BUCKET falls into its 1 / N of
 * the bucket range -- all windows, all points.  Every share then holds 1 / N of the mixed additions AND 1 / N of the buckets to merge
 * and fold (a row share repeats that tail in full on every rank), at the price of every rank scanning all digits and keeping all window
 * tables (no bbgpu_set_table_share).  Balanced for uniform digits; the partial sums of shares 0 .. N-1 add up to the MSM (bbgpu_g1_sum).
 * N <= the rows of the bucket matrix (256 at 17-bit windows).  Replaces the slicing of scalar_multiplication.cpp:703-738 (there: ranges of
 * POINTS per thread, summed at the end) at the multi-GPU level. */
int bbgpu_msm_g1_device_buckets_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int share, int share_count, void* hip_stream);
int bbgpu_srs_has_window_tables(int srs_handle); /* 1 / 0, < 0: unknown handle */
int bbgpu_msm_g1_wait(int ticket, uint64_t out[12]);
/* Whole-batch entry (the prover commits 3 / 1 / 3 / 2 polynomials per round over the same SRS,
 * prover.cpp:65-122,650-658): `jobs` (1..4) resident scalar vectors of n scalars each against points [offset, offset + n) of a
 * table registered WITH window tables, issued as ONE pass through the pipeline (per table segment touched) -- one bucket set per job in
 * the shared sort / accumulate / merge / reduction kernels -- so the batch pays one chain of launches and dependent additions, not `jobs`.
 * bbgpu_msm_g1_batch_wait writes jobs x 12 limbs (normalised).  Uses one of the eight tickets (and, over several segments, a free
 * one as its helper). */
int bbgpu_msm_g1_device_batch_async(int srs_handle, size_t offset, const uint64_t* const* d_scalars, int jobs, size_t n, void* hip_stream);
int bbgpu_msm_g1_batch_wait(int ticket, uint64_t* out);
/* out = sum of `count` normalised/Jacobian points (infinity flags honoured), normalised.  Host arithmetic. */
int bbgpu_g1_sum(const uint64_t* points12, size_t count, uint64_t out[12]);

/* ---- resident polynomial helpers -------------------------------------------------------------------
 * The O(n) loops of the prover that sit between the transforms and the commitments, on device-resident vectors in the
 * reference's memory format (n x 4 limbs, Montgomery; any representative below 2^256 in, canonical out).  All are
 * asynchronous on `hip_stream` (NULL = the legacy default stream) except where a host value is returned. */
/* polynomial_arithmetic::evaluate(coeffs, z, n) (polynomial_arithmetic.cpp:337-373): sum_i coeffs[i] z^i, canonical */
int bbgpu_fr_evaluate_device(const uint64_t* d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4], void* hip_stream);
/* fr::batch_invert(coeffs, n) (fields/field.hpp:503-522), in place; every element must be non-zero */
int bbgpu_fr_batch_invert_device(uint64_t* d_values, size_t n, void* hip_stream);
/* running products (the six accumulator chains of prover.cpp:194-202 are the exclusive prefix form):
 * out[i] = prod of in[j] over j < i (exclusive) or j <= i (inclusive); reverse != 0 scans from the top (j > i / j >= i) */
int bbgpu_fr_product_scan_device(const uint64_t* d_in, uint64_t* d_out, size_t n, int reverse, int inclusive, void* hip_stream);
/* polynomial_arithmetic::mul(a, b, r, domain) (:328-335) */
int bbgpu_fr_mul_device(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, void* hip_stream);
/* polynomial_arithmetic::compute_kate_opening_coefficients(src, dest, z, n) (:562-591): dest = (F(X) - F(z)) / (X - z),
 * returns F(z) in f_of_z (may be NULL).  dest may alias src only if it IS src. */
int bbgpu_kate_opening_device(const uint64_t* d_src, uint64_t* d_dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4], void* hip_stream);
/* polynomial_arithmetic::compute_lagrange_polynomial_fft(l_1, src_domain, target_domain) (:381-476): n_target values */
int bbgpu_lagrange_l1_fft_device(uint64_t* d_l_1, size_t n_src, size_t n_target, void* hip_stream);
/* polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial(coeffs, src_domain, target_domain) (:478-560), in place */
int bbgpu_divide_by_pseudo_vanishing_device(uint64_t* d_coeffs, size_t n_src, size_t n_target, void* hip_stream);
/* waffle::compute_permutation_lagrange_base_single(output, permutation, small_domain) (permutation.hpp:15-87) */
int bbgpu_permutation_lagrange_base_device(uint64_t* d_out, const uint32_t* d_mapping, size_t n, void* hip_stream);

/* The same on host buffers (copied to the device and back): what the C++ shim forwards the co-resident functions of the replaced
 * translation unit to, so that polynomial_arithmetic.o / scalar_multiplication.o can be left out of the link altogether. */
int bbgpu_fr_evaluate(const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4]);
int bbgpu_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4]);
int bbgpu_lagrange_l1_fft(uint64_t* l_1, size_t n_src, size_t n_target);
int bbgpu_divide_by_pseudo_vanishing(uint64_t* coeffs, size_t n_src, size_t n_target);
/* polynomial_arithmetic::get_lagrange_evaluations(z, domain) (:594-626): out = {Z_H*(z), L_1(z), L_{n-1}(z)}; host arithmetic */
int bbgpu_lagrange_evaluations(const uint64_t z[4], size_t n, uint64_t out[12]);
/* scalar_multiplication::generate_pippenger_point_table(points, table, n) (scalar_multiplication.cpp:131-140); points may alias table */
int bbgpu_generate_point_table(const uint64_t* points, uint64_t* table, size_t n);

/* ---- resident PLONK prover (BASELINE config 5) ------------------------------------------------------
 * waffle::Prover for the standard arithmetic circuit with every polynomial resident in HBM.  The circuit description is
 * the state StandardComposer::preprocess() hands the reference's Prover (standard_composer.cpp:163-220, prover.hpp:44-59,
 * widgets/arithmetic_widget.hpp:45-49): per-gate wire VALUES, the three sigma permutation mappings (low bits: gate index,
 * bits 30-31: 0 left / 1 right / 2 output wire) and the five selector VALUES, all of length n = 2^k.  The SRS is a registered
 * / generated handle holding at least n points (monomials[i] = x^i G, reference_string.cpp:16-35). */
typedef struct {
    size_t n;
    const uint64_t *w_l, *w_r, *w_o;                                   /* n x 4 limbs each */
    const uint32_t *sigma_1_mapping, *sigma_2_mapping, *sigma_3_mapping; /* n each */
    const uint64_t *q_m, *q_l, *q_r, *q_o, *q_c;                       /* n x 4 limbs each */
    /* optional second widget, all three or none (NULL): the boolean-constraint selectors BoolComposer::preprocess() hands
     * ProverBoolWidget (bool_composer.cpp:68-143, widgets/bool_widget.hpp): q_bl, q_br, q_bo, n x 4 limbs each */
    const uint64_t *q_bl, *q_br, *q_bo;
    /* or (not both) the MiMC widget's selectors as MiMCComposer::preprocess() hands ProverMiMCWidget (mimc_composer.cpp:170-250,
     * widgets/mimc_widget.hpp): q_mimc_selector, q_mimc_coefficient (the round constants), n x 4 limbs each; NULL without it.
     * The proof then also carries w_o_shifted_eval and q_mimc_coefficient_eval. */
    const uint64_t *q_mimc_selector, *q_mimc_coefficient;
    /* optional sequential widget (ExtendedComposer::preprocess(), extended_composer.cpp:460-607, widgets/sequential_widget.hpp): q_o_next,
     * the selector of the NEXT gate's output wire in the arithmetic identity, n x 4 limbs; NULL without it.  May be combined with the
     * bool widget (the ExtendedComposer's chain: arithmetic, sequential, bool), not with the MiMC widget.  The proof then carries
     * w_o_shifted_eval. */
    const uint64_t *q_o_next;
} bbgpu_plonk_circuit;
/* proof = waffle::plonk_proof (waffle_types.hpp:18-45) in its own field order: W_L, W_R, W_O, Z_1, T_LO, T_MID, T_HI,
 * PI_Z, PI_Z_OMEGA (affine x, y: 8 limbs each), then w_l_eval, w_r_eval, w_o_eval, sigma_1_eval, sigma_2_eval,
 * z_1_shifted_eval, linear_eval, and the widget-dependent w_l_shifted_eval, w_r_shifted_eval, w_o_shifted_eval, q_c_eval,
 * q_mimc_coefficient_eval (4 limbs each; zero unless a widget fills them: the MiMC widget sets w_o_shifted_eval and
 * q_mimc_coefficient_eval, the sequential widget w_o_shifted_eval); Montgomery form, canonical -- byte-identical to the reference's proof */
#define BBGPU_PLONK_PROOF_WORDS 120
int bbgpu_plonk_prover_create(const bbgpu_plonk_circuit* circuit, int srs_handle); /* returns a prover handle >= 0 */
int bbgpu_plonk_prover_set_witness(int prover, const uint64_t* w_l, const uint64_t* w_r, const uint64_t* w_o);
int bbgpu_plonk_construct_proof(int prover, uint64_t proof_out[BBGPU_PLONK_PROOF_WORDS]); /* Prover::construct_proof, prover.cpp:661-670 */
/* waffle::preprocess(prover) (preprocess.hpp:16-55, arithmetic_widget.cpp:128-157, bool_widget.cpp:118-152): the verification key of
 * the circuit -- SIGMA_1, SIGMA_2, SIGMA_3, the commitments to q_m, q_l, q_r, q_o, q_c, and with the bool widget those to q_bl, q_br,
 * q_bo, with the MiMC widget q_mimc_coefficient, q_mimc_selector, with the sequential widget q_o_next (before the bool widget's three) --
 * affine x, y: 8 limbs each; 8 to 12 points, pass room for 12 */
#define BBGPU_PLONK_VK_WORDS 96
int bbgpu_plonk_preprocess(int prover, uint64_t vk_out[BBGPU_PLONK_VK_WORDS]);
int bbgpu_plonk_last_challenges(int prover, uint64_t out[20]); /* beta, gamma, alpha, z, nu (waffle_types.hpp:9-16) */
int bbgpu_plonk_last_timing(int prover, double ms_out[4]);     /* construct_proof wall ms: total, in commitments, rest, first-use preparation */
int bbgpu_plonk_prover_destroy(int prover);
/* challenge.hpp:64-112 recomputed from a finished proof: gamma, beta, alpha, z (4 limbs each).  Host only, no GPU needed. */
int bbgpu_plonk_challenges_from_proof(const uint64_t proof[BBGPU_PLONK_PROOF_WORDS], uint64_t out[16]);

/* ---- host fallbacks of the drop-in boundary -----------------------------------------------------------------------
 * The reference API has no error channel (assert.hpp:13-23; batched_scalar_multiplications prints and returns, scalar_multiplication.cpp:680-684),
 * so a GPU call that fails at run time -- no device, an allocation refused on a shared GPU, a launch failure -- must not stop the prover:
 * shim/bb_shim.cpp logs the library's error once and computes the same result with these entries (csrc/host_fallback.hpp: textbook bucket
 * method / radix-2 transform / O(n) loops on the library's own host field code, a few host threads; never oracle/).  They make no HIP call,
 * take no lock and keep no state (re-entrant), accept what the GPU entries accept (any representative below 2^256) and return the same bytes
 * (canonical; MSM results normalised).  The GPU entries above never call them. */
int bbgpu_host_msm_g1(const uint64_t* scalars, const uint64_t* points, size_t n, int plain_table /* 0: 2n-entry endo table, 1: n-entry table */, uint64_t out[12]);
int bbgpu_host_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant);
int bbgpu_host_fr_evaluate(const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4]);
int bbgpu_host_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4]);
int bbgpu_host_lagrange_l1_fft(uint64_t* l_1, size_t n_src, size_t n_target);
int bbgpu_host_divide_by_pseudo_vanishing(uint64_t* coeffs, size_t n_src, size_t n_target);

/* ---- device self-test: known-answer entry points for the field and group layer ------------------------------------
 * One GPU lane per case runs the device arithmetic every kernel is built from (csrc/fe.hpp incl. the gfx950 asm products, csrc/g1.hpp);
 * operands and results in the reference's memory format, canonical.  What each op returns (a, b = the operands' residues):
 * field_impl_int128.tcc:72-137,149-263 / group.hpp:153-448 semantics. */
enum {
    BBGPU_SELFTEST_MUL = 0,        /* a b                         field::__mul          */
    BBGPU_SELFTEST_SQR = 1,        /* a^2                         field::__sqr          */
    BBGPU_SELFTEST_ADD = 2,        /* a + b                       field::__add          */
    BBGPU_SELFTEST_SUB = 3,        /* a - b                       field::__sub          */
    BBGPU_SELFTEST_NEG = 4,        /* -a                          field::__neg          */
    BBGPU_SELFTEST_MUL_ADD = 5,    /* a b + (a + b)(a - b)        two products, one Montgomery reduction */
    BBGPU_SELFTEST_MUL_SUB = 6,    /* a b - 2 a b                 the same with a negated operand */
    BBGPU_SELFTEST_LAZY_LIMBS = 7, /* 2a 3b                       unnormalised limbs at the multiplier's limit */
    BBGPU_SELFTEST_LAZY_WEAK = 8,  /* 4a (b - a)                  limbs beyond it: renormalised inside mul() */
    BBGPU_SELFTEST_LAZY_VALUE = 9, /* 28 a                        value bound at its maximum, 168 p < 2^261, through the multiplier */
    BBGPU_SELFTEST_REDUCE = 10,    /* 28 a                        the same through reduce_value() */
    BBGPU_SELFTEST_SQR_LAZY = 11,  /* (2a - b)^2 */
    BBGPU_SELFTEST_ZERO_TESTS = 12,/* limb 0: bit 0 = (a - b == 0), bit 1 = ((a - b) a == 0) */
    BBGPU_SELFTEST_MUL_ADDHI = 13, /* a b - a                     the product with a third operand added inside its reduction (in place; the mixed addition's P and R) */
    BBGPU_SELFTEST_SQR_ADDHI = 14  /* a^2 - (b + 2a)              the same for the squaring, addend with unnormalised limbs (the mixed addition's X3) */
};
enum {
    BBGPU_SELFTEST_G1_MADD = 0,      /* p + (q.x, q.y)                                             g1::mixed_add, group.hpp:219-322 */
    BBGPU_SELFTEST_G1_ADD = 1,       /* p + q                                                      g1::add, :324-448 */
    BBGPU_SELFTEST_G1_DBL = 2,       /* 2 p                                                        g1::dbl, :153-217 */
    BBGPU_SELFTEST_G1_DBL_AFFINE = 3,/* 2 p for an affine p (the P + P branch of the mixed addition) */
    BBGPU_SELFTEST_G1_MADD_NEG = 4,  /* p - (q.x, q.y): the conditionally negated operand the bucket accumulation feeds (group_impl_asm.tcc:71-153) */
    BBGPU_SELFTEST_G1_QUAD_ADD = 5   /* p + q by the four-lanes-per-point addition of the bucket reduction (csrc/g1_quad.hpp) */
};
/* field: 0 = fq, 1 = fr; a, b, out: n x 4 limbs */
int bbgpu_selftest_field(int field, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* p, q: n x 12 limbs (Jacobian {x, y, z}, infinity flag honoured); out: n x 16 limbs {X, Y, ZZ, ZZZ} with x = X / ZZ, y = Y / ZZZ,
 * ZZ = 0 for infinity (the kernels' extended-Jacobian form; the caller normalises) */
int bbgpu_selftest_g1(int op, const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out);

/* ---- instrumentation (bench.py) ---------------------------------------------------------------------------------
 * Device time in milliseconds of the kernels launched by the most recent bbgpu_*_device call on this thread, measured
 * with hipEvents on the stream the kernels ran on.  index: 0 = total, then per stage: 1 digits, 2 sort, 3 accumulation,
 * 4 merge, 5 row/column sums, 6 final sums, 7 accumulation without the time it sat queued behind the previous MSM's accumulation. */
int bbgpu_last_timing(float* ms_out, int max_entries);
void bbgpu_set_timing(int level); /* 0 off; 1 an event after every stage (adds ~0.08 ms of marker latency to a pipelined 2^20 step); 2 only
                                     the pair around the accumulation (indices 3 and 7 are filled): what bench.py's timed region uses */

#ifdef __cplusplus
}
#endif
#endif /* BBGPU_H */
