// bbgpu_internal.h -- declarations shared by the HIP translation units and the C-ABI shim (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/bbgpu.h"

namespace bbgpu {

void set_error(const char* fmt, ...);

// capi.hip: EVERY device allocation, host<->device copy of a caller's buffer and launch check of the library goes through these four, so that
// (i) a test can make the k-th one fail on a machine that HAS a GPU (BBGPU_FAIL_AT / bbgpu_fault_inject: "alloc:k", "h2d:k", "d2h:k", "launch:k" -- the
// error contract of the drop-in boundary, SURVEY 8b "Errors"; no kernel reads any of this) and (ii) the live device allocations are counted
// (bbgpu_fault_stats: a leak on an error path shows as allocations that outlive bbgpu_shutdown()).
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
hipError_t h2d_async(void* dst, const void* src, size_t bytes, hipStream_t st);
hipError_t d2h_async(void* dst, const void* src, size_t bytes, hipStream_t st);
hipError_t launch_check(); // hipGetLastError() after a launch (or a chain of launches)
// a device buffer that is freed on every way out of a function unless it was handed on (release())
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)dev_free(p); }
    template <class T> T* as() const { return static_cast<T*>(p); }
    template <class T> T* release() { T* r = static_cast<T*>(p); p = nullptr; return r; }
};
void fault_absorbed();     // a failure the library rode out by itself (an SRS kept without its window tables): counted, so a test can tell it from a lost injection

// ntt.hip
int ntt_device(uint64_t* d_coeffs, uint64_t* d_scratch, int log2n, int kind, const uint64_t* constant_m256, hipStream_t st);
int ntt_device_batch(uint64_t* d_coeffs, size_t stride_elems, int batch, uint64_t* d_scratch, int log2n, int kind, const uint64_t* constant_m256,
                     hipStream_t st);
void ntt_release_tables();
size_t ntt_table_bytes(size_t* cap_out, int* sets_out); // device bytes the cached twiddle / twist tables hold now, their budget, how many domain sizes

// msm.hip
namespace host { struct Xyzz; }
struct MsmTiming {
    int count = 0;
    float ms[8]; // [0] total device time, then digits, sort, accumulate kernel, merge, row/col folds, slices+collect,
                 // [7] accumulate kernel without the time it sat queued behind the previous MSM's accumulation (two MSMs in flight)
};
struct MsmWorkspace {
    uint8_t* base = nullptr;
    size_t cap = 0;
    void* h_out = nullptr; // pinned
    static size_t bytes_needed(size_t n, int c, int nw, int ng = 0); // nw (job, window) pairs, ng bucket sets (0: one per pair = no window tables)
    int ensure(size_t bytes);
    void release();
};
// One MSM of a slot may be issued as several PIECES -- point ranges against different segments of the window tables (an SRS above 2^20 points keeps
// one table per <= 2^20-point segment, capi.hip), enqueued back to back on the slot's stream and sharing its workspace in stream order; every piece
// writes its own 64-slot groups of the pinned result array and the host adds the piece sums (a sum over a range of points is a plain term of the MSM).
struct MsmPiece {
    uint32_t c = 0, nw = 0, wb = 0, hbits = 0, lbits = 0;
    uint32_t hout_group = 0; // first 64-slot group of this piece in ws.h_out
    bool trivial = false;
};
constexpr int MSM_MAX_JOBS = 4;        // MSMs over the same points issued as one batch (one bucket set each)
constexpr int MSM_MAX_PIECES = 64;      // per slot: an MSM without a free helper slot keeps all its pieces on one
constexpr uint32_t MSM_HOUT_GROUPS = 256; // 64-slot groups of the pinned result array (pieces x jobs, or the windows of one table-less piece): 2 MiB per slot
struct MsmSlot {
    MsmWorkspace ws;
    MsmPiece piece[MSM_MAX_PIECES];
    int npieces = 0;
    bool append = false; // set by the caller before an issue: add a piece to the MSM already issued on this slot instead of starting a new one
    int helper = -1;     // capi.hip: a second slot (own stream and workspace) that carries every other piece of this ticket, or -1
    bool is_helper = false; // this slot is pending as the helper of another ticket: collected through its owner only
    bool reserved = false;  // capi.hip: a synchronous entry point is cycling its jobs / ranges through this slot: not to be taken as a helper meanwhile
    hipStream_t stream = nullptr; // the slot's own stream (used when the caller passes none)
    hipEvent_t done = nullptr;
    hipEvent_t ev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    bool pending = false, trivial = false, timed = false, timed_light = false;
    size_t n = 0;
    uint32_t jobs = 1;
    uint64_t acc_seq = 0; // position of this MSM's accumulation in the process-wide sequence of timed accumulations (0 = none)
    // set by the caller before an issue: other MSMs are in flight, so this one is paced by the instructions it issues, not by its dependent chain
    // (msm_issue_batch then takes longer chunks and the two-step row / column sums)
    bool throughput = false;
    void release();
};
int msm_choose_c(size_t n);
int msm_num_windows(int c);
int msm_issue(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n, int wb,
              int we, hipStream_t st, int want_timing);
int msm_issue_batch(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* const* d_scalars_v, int jobs,
                    size_t n, int wb, int we, hipStream_t st, int want_timing, uint32_t row_i0 = 0, uint32_t row_i1 = 0xffffffffu, uint32_t brow0 = 0,
                    uint32_t brow1 = 0xffffffffu);
int msm_issue_buckets(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n,
                      uint32_t share, uint32_t share_count, hipStream_t st, int want_timing);
int msm_issue_rows(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n,
                   uint64_t row_begin, uint64_t row_end, hipStream_t st, int want_timing);
int msm_finish_batch(MsmSlot& S, host::Xyzz* results, MsmTiming* timing);
int srs_build_table(const uint32_t* d_srs, size_t n, int c, int num_windows, int w_begin, int w_end, uint32_t** d_alloc_out, uint32_t** d_tab_out, hipStream_t st);
int msm_finish(MsmSlot& S, host::Xyzz* result, MsmTiming* timing);
int srs_upload(const uint64_t* host_table, size_t n, uint32_t** d_srs_out, hipStream_t st, size_t stride_bytes = 128);
int srs_upload_into(const uint64_t* host_table, size_t n, uint32_t* d_raw, uint32_t* d_srs, hipStream_t st, size_t stride_bytes);
int srs_generate(const uint64_t* x_mont256, size_t first, size_t n, uint32_t** d_srs_out, uint64_t* host_table_out, hipStream_t st);

// capi.hip: the caller's host buffers cross the link through the library's OWN pinned buffers (see host_to_device)
int host_to_device(void* d_dst, const void* h_src, size_t bytes, hipStream_t st);
int device_to_host_sync(void* h_dst, const void* d_src, size_t bytes, hipStream_t st, bool* touched = nullptr); // *touched: h_dst may have been written (even on failure)
void host_stage_release();
int bind_calling_thread(); // initialises the library if need be and makes the bound device the calling thread's current one (HIP's current device is per thread)

// plonk.hip
std::mutex& plonk_mutex();        // taken BEFORE capi.hip's mutex wherever both are held
void plonk_release_all_locked();  // caller holds plonk_mutex()

} // namespace bbgpu
