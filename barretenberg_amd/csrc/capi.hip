// capi.hip -- the C-ABI of libbbgpu.so (include/bbgpu.h): context, SRS registry, host<->device staging.
// One process drives one GPU (bbgpu_init(device)); all entry points are serialised by one mutex, which also makes the
// reference's concurrent pippenger() calls from an OpenMP region (scalar_multiplication.cpp:731-738) safe.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <unordered_map>
#include <vector>

#include "bbgpu_internal.h"
#include "host_g1.hpp"
#include "host_g2.hpp"
#include "host_small.hpp"
#include "host_fallback.hpp"
#include "host_copy_pool.hpp"
#include "poly.h"

namespace bbgpu {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- the funnels of bbgpu_internal.h: allocation accounting + fault injection ---------------------------------------------------------------
// BBGPU_FAIL_AT (read once, when the first funnel is reached) or bbgpu_fault_inject(): "<kind>:<k>" with kind in alloc | h2d | d2h | launch makes the
// k-th call (k = 0: the next one) of that kind fail ONCE with the error a real failure of that kind returns, without touching the device; the
// call after it works again.  Counters restart whenever a spec is set.  Host-side only: no kernel reads any of it.
namespace {
enum FaultKind { F_ALLOC = 0, F_H2D = 1, F_D2H = 2, F_LAUNCH = 3, F_KINDS = 4 };
struct FaultState {
    std::mutex mu;
    std::atomic<int> armed_kind{ -1 };   // -1: nothing armed (the one load every funnel call pays)
    uint64_t fail_at = 0;
    std::atomic<uint64_t> calls[F_KINDS] = {};
    std::atomic<uint64_t> fired{ 0 }, absorbed{ 0 };
    std::unordered_map<void*, size_t> live; // device allocations of the library that are live now
    uint64_t live_bytes = 0;
    bool env_read = false;
};
FaultState& fault()
{
    static FaultState* f = new FaultState(); // never destroyed: funnels run during static destruction of other objects too
    return *f;
}
int fault_parse(const char* spec, int* kind, uint64_t* at)
{
    static const char* names[F_KINDS] = { "alloc", "h2d", "d2h", "launch" };
    if (!spec || !*spec) { *kind = -1; return 0; }
    for (int k = 0; k < F_KINDS; k++) {
        const size_t len = strlen(names[k]);
        if (!strncmp(spec, names[k], len) && spec[len] == ':') {
            char* end = nullptr;
            *at = strtoull(spec + len + 1, &end, 0);
            if (end == spec + len + 1 || *end) return -1;
            *kind = k;
            return 0;
        }
    }
    return -1;
}
int fault_set(const char* spec)
{
    FaultState& F = fault();
    int kind = -1;
    uint64_t at = 0;
    if (fault_parse(spec, &kind, &at)) return -1;
    std::lock_guard<std::mutex> lk(F.mu);
    F.env_read = true; // an explicit spec overrides the environment
    for (auto& c : F.calls) c.store(0);
    F.fired.store(0);
    F.absorbed.store(0);
    F.fail_at = at;
    F.armed_kind.store(kind);
    return 0;
}
// true: this call is the one to fail
bool fault_hit(FaultKind kind)
{
    FaultState& F = fault();
    if (!F.env_read) {
        std::lock_guard<std::mutex> lk(F.mu);
        if (!F.env_read) {
            F.env_read = true;
            int k = -1;
            uint64_t at = 0;
            const char* e = getenv("BBGPU_FAIL_AT"); // testing: "alloc:k" / "h2d:k" / "d2h:k" / "launch:k" makes the k-th such call of the process fail once (include/bbgpu.h, bbgpu_fault_inject)
            if (e && fault_parse(e, &k, &at) == 0 && k >= 0) {
                F.fail_at = at;
                F.armed_kind.store(k);
            } else if (e && *e) {
                fprintf(stderr, "bbgpu: BBGPU_FAIL_AT=%s not understood (alloc:k | h2d:k | d2h:k | launch:k)\n", e);
            }
        }
    }
    const uint64_t n = F.calls[kind].fetch_add(1);
    if (F.armed_kind.load(std::memory_order_relaxed) != (int)kind) return false;
    std::lock_guard<std::mutex> lk(F.mu);
    if (F.armed_kind.load() != (int)kind || n != F.fail_at) return false;
    F.armed_kind.store(-1); // one shot
    F.fired.fetch_add(1);
    return true;
}
} // namespace

hipError_t dev_malloc(void** p, size_t bytes)
{
    if (fault_hit(F_ALLOC)) {
        *p = nullptr;
        set_error("hipMalloc(%zu bytes) -> %s [injected by BBGPU_FAIL_AT]", bytes, hipGetErrorString(hipErrorOutOfMemory));
        return hipErrorOutOfMemory;
    }
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && *p) {
        FaultState& F = fault();
        std::lock_guard<std::mutex> lk(F.mu);
        F.live[*p] = bytes;
        F.live_bytes += bytes;
    }
    return e;
}
hipError_t dev_free(void* p)
{
    if (!p) return hipSuccess;
    {
        FaultState& F = fault();
        std::lock_guard<std::mutex> lk(F.mu);
        auto it = F.live.find(p);
        if (it != F.live.end()) {
            F.live_bytes -= it->second;
            F.live.erase(it);
        }
    }
    return hipFree(p);
}
hipError_t h2d_async(void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (fault_hit(F_H2D)) {
        set_error("hipMemcpyAsync(host to device, %zu bytes) -> %s [injected by BBGPU_FAIL_AT]", bytes, hipGetErrorString(hipErrorInvalidValue));
        return hipErrorInvalidValue;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
}
hipError_t d2h_async(void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (fault_hit(F_D2H)) {
        set_error("hipMemcpyAsync(device to host, %zu bytes) -> %s [injected by BBGPU_FAIL_AT]", bytes, hipGetErrorString(hipErrorInvalidValue));
        return hipErrorInvalidValue;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
}
hipError_t launch_check()
{
    const hipError_t e = hipGetLastError(); // a real failure is reported (and cleared) first
    if (e != hipSuccess) return e;
    if (fault_hit(F_LAUNCH)) {
        set_error("kernel launch -> %s [injected by BBGPU_FAIL_AT]", hipGetErrorString(hipErrorLaunchFailure));
        return hipErrorLaunchFailure;
    }
    return hipSuccess;
}
void fault_absorbed() { fault().absorbed.fetch_add(1); }

static host::CopyPool g_copy_pool; // CPU copies into / out of the pinned staging buffers (host_copy_pool.hpp); also hashes point tables (exact cache mode)

namespace {

struct SrsEntry {
    const uint64_t* host_ptr; // may be null for device-generated tables
    size_t n;
    uint32_t* d_srs;
    bool live;
    // Pre-shifted window tables, one per SEGMENT of at most 2^24 / W points (the sorted entries carry a 24-bit table row: 2^20 points at 15 windows).
    // A larger SRS keeps several segments of equal length and an MSM over it runs as one piece per segment it touches, the piece sums added on
    // the host -- the point-range split of scalar_multiplication.cpp:703-738 inside one GPU.  Empty: no tables (per-window bucket sets).
    struct TabSeg {
        size_t first = 0, n = 0;          // points [first, first + n) of the entry
        uint32_t* d_tab = nullptr;       // [tab_W][n] rows of 64 bytes: the address window 0 has (or would have)
        uint32_t* d_tab_alloc = nullptr; // the allocation itself: starts at window tab_wb when only a share of the windows is kept
    };
    std::vector<TabSeg> segs;
    bool has_tab() const { return !segs.empty(); }
    int tab_c = 0, tab_W = 0, tab_wb = 0, tab_we = 0; // windows [tab_wb, tab_we) are resident (the same for every segment)
    // Address-keyed lookups are only trusted after a CONTENT check: one 64-bit hash per base point (the even table entry the kernels
    // read), taken when the table was uploaded.  A lookup re-hashes the first, the last and up to 14 evenly spaced rows of the range
    // the caller passed (it never touches host memory outside that range: the old table may have been freed) and compares.
    std::vector<uint64_t> row_hash;
    bool auto_registered = false; // created by a host-pointer MSM on first sight: evictable (stale contents, overlap, LRU under the byte cap)
    uint64_t last_use = 0;
    size_t bytes = 0; // device bytes held (points + window tables)
    bool handle_exposed = false; // the index was returned to a caller as a handle: the slot is never reused for another table
    uint64_t check_phase = 0;    // rotates the rows a content check samples (contents_match)
    bool stale_for_host = false; // exact mode found the caller's memory changed under an EXPLICITLY registered table: host-pointer calls no longer use it (the handle stays valid)
    bool validate_full = false;  // EXACT mode (bbgpu_srs_set_validate / BBGPU_SRS_VALIDATE=full): every host-pointer MSM re-hashes the whole range it uses
};

// eight independent multiplications (odd multipliers: a change of one limb always changes the sum) instead of a dependent chain: the full content
// check of BBGPU_SRS_VALIDATE=full hashes every row of the range on every call and must run at memory speed
inline uint64_t hash_row(const uint64_t* row8)
{
    static const uint64_t M[8] = { 0x9e3779b97f4a7c15ULL, 0xbf58476d1ce4e5b9ULL, 0x94d049bb133111ebULL, 0xd6e8feb86659fd93ULL,
                                   0xca5a826395121157ULL, 0xff51afd7ed558ccdULL, 0xc4ceb9fe1a85ec53ULL, 0x2545f4914f6cdd1dULL };
    uint64_t h = 0;
    for (int i = 0; i < 8; i++) h += (row8[i] ^ M[(i + 3) & 7]) * M[i];
    h ^= h >> 31;
    return h * 0x9e3779b97f4a7c15ULL;
}

struct Context {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<SrsEntry> srs;
    static constexpr int NSLOT = 8; // asynchronous MSMs in flight (each with its own workspace and stream; workspaces are allocated on first use)
    MsmSlot slot[NSLOT];
    int next_slot = 0;
    uint64_t* d_stage = nullptr; // scalars / coefficients staging
    size_t stage_cap = 0;
    uint64_t* d_stage2 = nullptr; // second scalar staging buffer (pipelined batches)
    size_t stage2_cap = 0;
    uint64_t* d_scratch = nullptr; // NTT scratch
    size_t scratch_cap = 0;
    uint64_t* d_small_tab = nullptr; // a small point table that is used once (raw upload + working form): kept, so that such a call allocates and frees nothing
    size_t small_tab_cap = 0;
    int timing = 0; // 0 off, 1 every stage, 2 the accumulation only (bbgpu_set_timing)
    bool precompute = true; // build window tables for registered SRS (bbgpu_set_precompute)
    uint64_t use_clock = 0;  // LRU clock of the SRS cache
    size_t srs_cache_cap = (size_t)16 << 30; // device bytes the auto-registered tables may hold together (BBGPU_SRS_CACHE_BYTES)
    bool srs_validate_full = false;          // default of SrsEntry::validate_full for tables registered from now on
    // SURVEY 8b "small sizes": host-pointer MSMs of at most host_msm_max points against tables that are not resident, and host-buffer
    // transforms of at most host_ntt_max elements, are answered on the host (host_small.hpp); bbgpu_set_host_thresholds / BBGPU_HOST_MSM_MAX /
    // BBGPU_HOST_NTT_MAX.  Defaults from tools/small_sizes.py on MI355X + EPYC 9575F: MSM n = 4 / 20 / 32 / 64 host 0.16 / 0.31 / 0.41 /
    // 0.73 ms against 0.35-0.37 ms through the kernels (crossover near 24 points); transforms n = 4 / 16 / 64 host 9 / 13 / 68 us against 37-43 us
    int host_msm_max = 24;
    int host_ntt_max = 16;
    bool host_env_read = false;
    int share_rank = 0, share_world = 1; // window share of the tables built from now on (bbgpu_set_table_share)
    int point_world = 1;                 // tables built from now on hold 1 / point_world of the points of a larger MSM (bbgpu_set_point_share)
    hipEvent_t helper_dep[NSLOT] = {};   // orders a helper slot's stream behind the caller's stream (issue_ticket)
    // Workspaces shared by every caller (NTT scratch, polynomial temporaries): users on different streams are chained by this event
    hipEvent_t shared_done = nullptr;
    hipStream_t shared_last = nullptr;
    bool shared_used = false;
    poly::Scratch poly_scratch; // workspace of the resident polynomial helpers
    uint64_t* d_poly_tmp = nullptr;
    size_t poly_tmp_cap = 0;
    MsmTiming last;
    // pinned staging for the callers' host buffers (host_to_device / device_to_host_sync)
    // 8 x 1 MiB (round 3: 2 x 4 MiB): the first chunk reaches the link after 1 MiB of CPU copy instead of 4, and up to seven chunks are queued behind
    // it -- bbgpu_ntt at 2^18 (8 MiB each way) 0.651 -> 0.617 ms, the reference prover on the shim level within its own noise (tools/stage_chunk_ab.sh)
    static constexpr size_t HOST_CHUNK = (size_t)1 << 20; // bytes per pinned staging buffer
    static constexpr int HOST_RING = 8;                   // staging buffers (a ring: the CPU copies run ahead of the DMA of the chunks before them)
    size_t host_chunk = HOST_CHUNK;                       // bytes of a buffer actually used per chunk (BBGPU_STAGE_CHUNK_BYTES <= HOST_CHUNK)
    void* h_stage[HOST_RING] = {};
    hipEvent_t h_stage_free[HOST_RING] = {};
    unsigned h_stage_next = 0;
    size_t host_stage_max = (size_t)8 << 20; // BBGPU_STAGE_MAX_BYTES: larger buffers are handed to hipMemcpyAsync as they are
};

std::recursive_mutex g_mu;
Context g_ctx;

#define CHK(x)                                                                                                         \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));                                \
            return BBGPU_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)

void read_host_env()
{
    // the staging knobs are read once per process whatever else happened; the two thresholds only while bbgpu_set_host_thresholds() has not set them
    // (round 5: that call used to switch off the reading of ALL four variables)
    static bool staging_read = false;
    const bool thresholds = !g_ctx.host_env_read;
    g_ctx.host_env_read = true;
    if (thresholds) {
        if (const char* e = getenv("BBGPU_HOST_MSM_MAX")) g_ctx.host_msm_max = atoi(e);
        if (const char* e = getenv("BBGPU_HOST_NTT_MAX")) g_ctx.host_ntt_max = std::min(64, atoi(e));
    }
    if (staging_read) return;
    staging_read = true;
    if (const char* e = getenv("BBGPU_STAGE_MAX_BYTES")) g_ctx.host_stage_max = (size_t)strtoull(e, nullptr, 0);
    if (const char* e = getenv("BBGPU_STAGE_CHUNK_BYTES")) g_ctx.host_chunk = std::min(Context::HOST_CHUNK, std::max((size_t)64 << 10, (size_t)strtoull(e, nullptr, 0))); // tuning knob
}

int ensure_init()
{
    if (g_ctx.ready) {
        // HIP's current device is PER THREAD (device 0 in a fresh one): a caller's worker thread -- the reference's OpenMP threads around pippenger(),
        // bench.py's issuing thread on rank r > 0 -- must allocate and launch on the device this process is bound to, not on device 0
        static thread_local int bound_device = -1;
        if (bound_device != g_ctx.device) {
            CHK(hipSetDevice(g_ctx.device));
            bound_device = g_ctx.device;
        }
        return BBGPU_OK;
    }
    // The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels of two streams that
    // share a queue run one after the other.  With the caller's streams beside them the four MSM slot streams landed on TWO queues: "three in
    // flight" was slower than two for that reason alone (rocprofv3 timeline, DESIGN_HISTORY.md 6; 2^16-point MSMs three in flight 0.168 -> 0.127 ms per
    // MSM with 8 queues, a 1/8 share four in flight 0.232 -> 0.205).  16: the eight slot streams, the library's own and the caller's.  Only effective when this is the process's first HIP call; a host
    // program that initialises HIP earlier sets the variable itself (bench.py and the Python binding do).
    (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
    // asked once per process: without a device every call would repeat the runtime's probe of the machine (~10 ms each; the shim's host
    // answers would be paced by it)
    static const int cnt = [] { int c = 0; return hipGetDeviceCount(&c) == hipSuccess ? c : 0; }();
    if (cnt == 0) {
        set_error("no HIP device available: the GPU entry points of libbbgpu have no CPU fallback");
        return BBGPU_ERR_HIP;
    }
    CHK(hipSetDevice(g_ctx.device));
    CHK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    // the slot streams are made HERE, one after the other: the runtime deals streams to hardware queues in creation order, and a slot
    // stream created later (first use of a third slot, in a process that has made other streams meanwhile) can land on the queue of another slot
    for (auto& sl : g_ctx.slot)
        if (!sl.stream) CHK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    CHK(hipEventCreateWithFlags(&g_ctx.shared_done, hipEventDisableTiming));
    g_ctx.shared_used = false;
    if (const char* e = getenv("BBGPU_SRS_CACHE_BYTES")) g_ctx.srs_cache_cap = (size_t)strtoull(e, nullptr, 0);
    if (const char* e = getenv("BBGPU_SRS_VALIDATE")) g_ctx.srs_validate_full = !strcmp(e, "full"); // full: host-pointer MSMs re-hash every row they use on every call (exact, ~+0.1 ms per 2^16 points hidden behind the kernels); default: 16 sampled rows
    g_ctx.ready = true;
    return BBGPU_OK;
}

// The NTT scratch and the polynomial temporaries are one set of buffers for all callers.  A call on stream `st` first waits (on the
// device) for the last user if that was another stream, and leaves its own completion event behind: two transforms issued back to
// back on two streams then run one after the other instead of overwriting each other's intermediate data.
int shared_begin(hipStream_t st)
{
    if (g_ctx.shared_used && g_ctx.shared_last != st) CHK(hipStreamWaitEvent(st, g_ctx.shared_done, 0));
    return BBGPU_OK;
}
int shared_end(hipStream_t st)
{
    CHK(hipEventRecord(g_ctx.shared_done, st));
    g_ctx.shared_last = st;
    g_ctx.shared_used = true;
    return BBGPU_OK;
}

int grow(uint64_t** buf, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return BBGPU_OK;
    if (*buf) (void)dev_free(*buf);
    *buf = nullptr;
    *cap = 0;
    CHK(dev_malloc((void**)buf, bytes));
    *cap = bytes;
    return BBGPU_OK;
}

} // namespace

// ---- the callers' host buffers ------------------------------------------------------------------------------------------------------
// A pageable buffer handed to hipMemcpyAsync is pinned in place by the runtime (a userptr mapping it keeps for later copies).  That is
// the fastest path while the caller keeps its buffers -- and a trap when it does not: the reference's Prover allocates its polynomials
// per proof, and when such a pinned range is unmapped the driver quiesces and later restores the process's GPU queues.  Seen from the
// unmodified reference prover linked on the shim (tools/shim_profile.py, BBGPU_TRACE_SRS=1): from the second proof of a process on, one
// hipMemcpyAsync per proof BLOCKED for 6-23 ms (a 2 MiB upload; of a 64 ms proof).  So buffers up to `host_stage_max` (8 MiB: every
// polynomial of a 2^16-gate proof, its 4n-coset vectors and its 8 MiB point table) cross through two pinned 4 MiB buffers of the
// library's own: CPU memcpy (30-50 GB/s on the boxes' EPYC 9575F), DMA from / to pinned memory, the copy of chunk k+1 under the DMA of
// chunk k.  Larger buffers keep the direct path: there a copy is 0.6 ms per 32 MiB against ~1 ms through one staging thread, and the
// stall is small against the work (profiles/r03_pcie.txt).

int bind_calling_thread()
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return ensure_init();
}
static int host_stage_ensure()
{
    for (int k = 0; k < Context::HOST_RING; k++)
        if (!g_ctx.h_stage[k]) {
            CHK(hipHostMalloc(&g_ctx.h_stage[k], Context::HOST_CHUNK, hipHostMallocDefault));
            CHK(hipEventCreateWithFlags(&g_ctx.h_stage_free[k], hipEventDisableTiming));
        }
    return BBGPU_OK;
}
// Both directions take the library mutex themselves (recursive: the capi entry points already hold it): the resident prover's uploads
// (plonk.hip, which holds only its own mutex) would otherwise race with a transform or an MSM of another thread on the staging buffers,
// their events and the single-producer copy pool.  Lock order everywhere: the prover's mutex first, then this one.
int host_to_device(void* d_dst, const void* h_src, size_t bytes, hipStream_t st)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    read_host_env();
    if (bytes == 0) return BBGPU_OK;
    if (bytes > g_ctx.host_stage_max) {
        CHK(h2d_async(d_dst, h_src, bytes, st));
        return BBGPU_OK;
    }
    if (int rc = host_stage_ensure()) return rc;
    const size_t CH = g_ctx.host_chunk;
    for (size_t off = 0; off < bytes; off += CH) {
        const size_t len = std::min(CH, bytes - off);
        const int k = (int)(g_ctx.h_stage_next++ % Context::HOST_RING);
        CHK(hipEventSynchronize(g_ctx.h_stage_free[k])); // the DMA that last read this buffer has finished (no-op before its first use)
        g_copy_pool.copy(g_ctx.h_stage[k], (const char*)h_src + off, len);
        CHK(h2d_async((char*)d_dst + off, g_ctx.h_stage[k], len, st));
        CHK(hipEventRecord(g_ctx.h_stage_free[k], st));
    }
    return BBGPU_OK;
}
// device -> caller's buffer, complete on return (everything enqueued on `st` before it has run as well).
// *touched (optional): set when the call may have written ANY byte of h_dst -- on a failure that tells an in-place caller whether its input is still
// whole (nothing touched: an ordinary error the shim answers on the host) or gone (BBGPU_ERR_LOST).
int device_to_host_sync(void* h_dst, const void* d_src, size_t bytes, hipStream_t st, bool* touched)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    read_host_env();
    if (touched) *touched = false;
    if (bytes > g_ctx.host_stage_max) {
        CHK(d2h_async(h_dst, d_src, bytes, st));
        if (touched) *touched = true; // the DMA engine owns the destination from here on
        CHK(hipStreamSynchronize(st));
        return BBGPU_OK;
    }
    if (int rc = host_stage_ensure()) return rc;
    // up to HOST_RING - 1 chunks are on the link (or queued for it) while one is copied out of its pinned buffer
    const size_t CH = g_ctx.host_chunk;
    const size_t chunks = (bytes + CH - 1) / CH;
    constexpr size_t AHEAD = Context::HOST_RING - 1;
    int kbuf[Context::HOST_RING] = {};
    auto enqueue = [&](size_t c) -> int {
        const int k = (int)(g_ctx.h_stage_next++ % Context::HOST_RING);
        kbuf[c % Context::HOST_RING] = k;
        CHK(hipEventSynchronize(g_ctx.h_stage_free[k]));
        CHK(d2h_async(g_ctx.h_stage[k], (const char*)d_src + c * CH, std::min(CH, bytes - c * CH), st));
        CHK(hipEventRecord(g_ctx.h_stage_free[k], st));
        return BBGPU_OK;
    };
    if (chunks == 0) {
        CHK(hipStreamSynchronize(st));
        return BBGPU_OK;
    }
    size_t queued = 0;
    for (; queued < chunks && queued < AHEAD; queued++)
        if (int rc = enqueue(queued)) return rc;
    for (size_t c = 0; c < chunks; c++) {
        if (queued < chunks) {
            if (int rc = enqueue(queued)) return rc;
            queued++;
        }
        const int k = kbuf[c % Context::HOST_RING];
        CHK(hipEventSynchronize(g_ctx.h_stage_free[k]));
        if (touched) *touched = true;
        g_copy_pool.copy((char*)h_dst + c * CH, g_ctx.h_stage[k], std::min(CH, bytes - c * CH));
    }
    return BBGPU_OK;
}
void host_stage_release()
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_copy_pool.shutdown();
    for (int k = 0; k < Context::HOST_RING; k++) {
        if (g_ctx.h_stage[k]) (void)hipHostFree(g_ctx.h_stage[k]);
        if (g_ctx.h_stage_free[k]) (void)hipEventDestroy(g_ctx.h_stage_free[k]);
        g_ctx.h_stage[k] = nullptr;
        g_ctx.h_stage_free[k] = nullptr;
    }
}

namespace {

constexpr size_t AUTO_REGISTER_MIN_POINTS = 1024; // host-pointer MSMs against unknown tables below this size do not cache the table

// BBGPU_TRACE_SRS=1: one line on stderr per event of the address-keyed point-table cache (register, evict, content mismatch)
bool trace_srs()
{
    static const bool on = getenv("BBGPU_TRACE_SRS") != nullptr;
    return on;
}
void free_entry(SrsEntry& e)
{
    if (trace_srs()) fprintf(stderr, "bbgpu srs: evict %p n=%zu auto=%d\n", (const void*)e.host_ptr, e.n, (int)e.auto_registered);
    // an asynchronous MSM may still be reading the table: drain the device first (rare path)
    (void)hipDeviceSynchronize();
    if (e.d_srs) (void)dev_free(e.d_srs);
    for (auto& sg : e.segs)
        if (sg.d_tab_alloc) (void)dev_free(sg.d_tab_alloc);
    e.segs.clear();
    e.d_srs = nullptr;
    e.live = false;
    e.row_hash.clear();
    e.row_hash.shrink_to_fit();
}
bool ranges_overlap(const SrsEntry& e, const uint64_t* p, size_t n)
{
    const uint8_t *a0 = (const uint8_t*)e.host_ptr, *a1 = a0 + e.n * 128, *b0 = (const uint8_t*)p, *b1 = b0 + n * 128;
    return a0 < b1 && b0 < a1;
}

// registers resident points; builds the pre-shifted window tables when enabled and the 24-bit row index allows it.
// An auto-registered table first evicts the auto-registered tables it overlaps in host memory (they are what used to live there) and
// the least recently used ones beyond the byte cap.
int add_srs(const uint64_t* host_ptr, size_t n, uint32_t* d_srs, bool auto_registered)
{
    if (trace_srs()) fprintf(stderr, "bbgpu srs: register %p n=%zu auto=%d\n", (const void*)host_ptr, n, (int)auto_registered);
    SrsEntry e;
    e.host_ptr = host_ptr;
    e.n = n;
    e.d_srs = d_srs;
    e.live = true;
    e.auto_registered = auto_registered;
    e.validate_full = g_ctx.srs_validate_full;
    e.last_use = ++g_ctx.use_clock;
    if (host_ptr) {
        e.row_hash.resize(n);
        struct Job { const uint64_t* p; uint64_t* h; } job{ host_ptr, e.row_hash.data() };
        g_copy_pool.for_range(n, (size_t)1 << 14, [](void* c, size_t lo, size_t hi) {
            const Job* j = static_cast<const Job*>(c);
            for (size_t i = lo; i < hi; i++) j->h[i] = hash_row(j->p + i * 16);
        }, &job);
    }
    // a rank's slice of a point-range split takes the window size of the whole MSM: measured on 1/4 and 1/8 slices of 2^20 points, four in flight,
    // 15-bit windows 0.356 / 0.189 ms per step, 17-bit 0.334 / 0.188 (16-bit at 1/8: 0.182) -- tools/slice_ab.py
    const size_t n_for_c = n * (size_t)g_ctx.point_world;
    int c = msm_choose_c(n_for_c);
    // with tables every window feeds one shared bucket set, so wider windows only cost bucket-reduction depth while each one
    // saved is n fewer mixed additions: measured on the resident prover (tools/plonk_bench.py), 2^16 gates 3.58 ms at c = 12,
    // 3.40 / 3.54 / 3.36 / 3.40 at c = 13 / 14 / 15 / 16; 2^18 gates 7.51 ms at c = 14, 7.19 at c = 15, 7.20 at c = 16
    if (n_for_c >= ((size_t)1 << 16) && c < 15) c = 15;
    // 17-bit windows (15 instead of 16 of them, signed digits up to +-2^16 kept as uint16 magnitude + sign bit, 2^16 buckets): one n-th fewer mixed additions.
    // Measured: single 2^20 MSM 1.611 -> 1.546 ms, two in flight 1.345 -> 1.287 ms/step (-4.3 %); prover 2^19 gates 11.89 -> 11.51 ms,
    // 2^20 gates 22.1-22.8 -> 22.0 ms; 2^18 gates unchanged (6.8 ms), so smaller tables keep c = 15 (measured again at the end of round 3, tools/plonk_bench.py:
    // 2^17 gates 3.51 / 3.40-3.50 / 3.49-3.52 ms at c = 15 / 16 / 17, 2^18 gates 5.74-5.80 / 5.58-5.87 / 5.58-5.63: within 3 %, and the reference fixtures sit around the 2^19 switch)
    if (n_for_c >= ((size_t)1 << 19)) c = 17;
    // ... but a slice of fewer than 2^18 points pays the row / column sums over 2^16 buckets for ~30 entries per bucket: 16-bit windows (2^15 buckets, one window
    // more) measured 0.179 against 0.182 ms per step at 2^17 points, four in flight, three alternating runs (15-bit windows: 0.189)
    if (g_ctx.point_world > 1 && c == 17 && n < ((size_t)1 << 18)) c = 16;
    if (const char* ev = getenv("BBGPU_TABLE_C")) c = std::min(17, std::max(4, atoi(ev))); // tuning knob: window size of the tables
    const int W = msm_num_windows(c);
    // segments: as few as the 24-bit row index allows, equal lengths (multiples of 8: the sort reads eight digits per load).  One up to 2^20 points;
    // beyond that the tables are kept up to BBGPU_TABLE_MAX_BYTES (default 64 GiB = 2^26 points) -- larger tables fall back to per-window bucket sets
    static const uint64_t tab_max_bytes = [] { const char* v = getenv("BBGPU_TABLE_MAX_BYTES"); return v ? strtoull(v, nullptr, 0) : (uint64_t)64 << 30; }();
    size_t seg_cap = (size_t)((((uint64_t)1 << 24) / (uint64_t)W) & ~(uint64_t)7);
    // testing knob: smaller segments, so that the piece machinery of the large MSMs can be driven against the oracle at sizes the oracle finishes in seconds
    // (read at registration; the decomposition changes, the sum does not)
    if (const char* v = getenv("BBGPU_TABLE_SEG_POINTS")) seg_cap = std::min(seg_cap, std::max<size_t>(64, (size_t)strtoull(v, nullptr, 0) & ~(size_t)7));
    const size_t nseg = (n + seg_cap - 1) / seg_cap;
    const size_t seg_n = nseg <= 1 ? n : ((((n + nseg - 1) / nseg) + 7) & ~(size_t)7);
    const bool want_tab = g_ctx.precompute && n >= 1024 && (nseg == 1 || ((uint64_t)n * W * 64 <= tab_max_bytes && nseg <= (size_t)MSM_MAX_PIECES));
    // a rank of an N-way row split touches windows [floor(W r / N), ceil(W (r + 1) / N)) only (bbgpu_set_table_share)
    const int twb = (int)((int64_t)W * g_ctx.share_rank / g_ctx.share_world);
    const int twe = (int)(((int64_t)W * (g_ctx.share_rank + 1) + g_ctx.share_world - 1) / g_ctx.share_world);
    e.bytes = n * 64 + (want_tab ? (size_t)(twe - twb) * n * 64 : 0);
    if (auto_registered) {
        for (auto& o : g_ctx.srs)
            if (o.live && o.auto_registered && o.host_ptr && host_ptr && ranges_overlap(o, host_ptr, n)) free_entry(o);
        for (;;) {
            size_t held = 0;
            SrsEntry* lru = nullptr;
            for (auto& o : g_ctx.srs)
                if (o.live && o.auto_registered) {
                    held += o.bytes;
                    if (!lru || o.last_use < lru->last_use) lru = &o;
                }
            if (!lru || held + e.bytes <= g_ctx.srs_cache_cap) break;
            free_entry(*lru);
        }
    }
    if (want_tab) {
        for (size_t first = 0; first < n; first += seg_n) {
            SrsEntry::TabSeg sg;
            sg.first = first;
            sg.n = std::min(seg_n, n - first);
            int rc = srs_build_table(d_srs + first * 16, sg.n, c, W, twb, twe, &sg.d_tab_alloc, &sg.d_tab, g_ctx.stream);
            if (rc) {
                // no room for the window tables (a shared GPU): the points stay resident and the MSMs over them take one bucket set per window --
                // slower (1.8 ms instead of 1.14 at 2^20) but on the GPU, instead of failing the registration and sending the caller to the host
                for (auto& o : e.segs) (void)dev_free(o.d_tab_alloc);
                e.segs.clear();
                (void)hipGetLastError();
                fault_absorbed();
                fprintf(stderr, "bbgpu: window tables of %zu bytes for an SRS of %zu points could not be built (%s): continuing without them\n",
                        (size_t)(twe - twb) * n * 64, n, g_err);
                g_err[0] = 0;
                break;
            }
            e.segs.push_back(sg);
        }
        if (e.segs.empty()) {
            e.bytes = n * 64;
        } else {
            e.tab_c = c;
            e.tab_W = W;
            e.tab_wb = twb;
            e.tab_we = twe;
        }
    }
    e.handle_exposed = !auto_registered;
    // a long-lived process that keeps re-registering tables on first sight must not grow the registry by one entry per eviction: dead slots
    // whose index no caller ever held are taken again
    if (auto_registered)
        for (size_t k = 0; k < g_ctx.srs.size(); k++)
            if (!g_ctx.srs[k].live && !g_ctx.srs[k].handle_exposed) {
                g_ctx.srs[k] = std::move(e);
                return (int)k;
            }
    g_ctx.srs.push_back(std::move(e));
    return (int)g_ctx.srs.size() - 1;
}
int entry_windows(const SrsEntry& e, size_t n)
{
    return e.has_tab() ? e.tab_W : msm_num_windows(msm_choose_c(n ? n : 1));
}
bool windows_resident(const SrsEntry& e, int wb, int we)
{
    if (!e.has_tab() || (wb >= e.tab_wb && we <= e.tab_we)) return true;
    set_error("windows [%d, %d) requested, this table keeps [%d, %d) of %d (bbgpu_set_table_share)", wb, we, e.tab_wb, e.tab_we, e.tab_W);
    return false;
}
// A free slot for an asynchronous MSM: slots 0 / 1 alternate for consecutive calls (two large MSMs in flight is the measured optimum for the
// full-size pipeline), the others take whatever else is in flight (shares of a split MSM, small MSMs: up to eight).  -1: all busy.
int pick_slot()
{
    int order[Context::NSLOT] = { g_ctx.next_slot, g_ctx.next_slot ^ 1 };
    for (int k = 2; k < Context::NSLOT; k++) order[k] = k;
    for (int k = 0; k < Context::NSLOT; k++)
        if (!g_ctx.slot[order[k]].pending) return order[k];
    set_error("all %d MSM slots are in flight: call bbgpu_msm_g1_wait first", Context::NSLOT);
    return -1;
}
// up to `want` slots that are not in flight, for the synchronous (host-pointer) entry points: they take what is free instead of insisting on
// slots 0 / 1, so a caller that holds asynchronous tickets -- or other threads doing so -- never makes pippenger() fail (the reference calls it
// from inside an OpenMP region, scalar_multiplication.cpp:731-738; calls are serialised by the library mutex, not refused)
int free_slots(int* out, int want)
{
    int got = 0;
    for (int k = 0; k < Context::NSLOT && got < want; k++)
        if (!g_ctx.slot[k].pending) out[got++] = k;
    return got;
}
// the slots a synchronous entry point cycles its jobs / ranges through: marked for the duration of the call, so that a multi-piece job on one of
// them does not take the other as its helper (it is free NOW and about to carry the next job)
struct SlotReservation {
    int a, b;
    SlotReservation(const int* sl, int n) : a(n > 0 ? sl[0] : -1), b(n > 1 ? sl[1] : -1)
    {
        if (a >= 0) g_ctx.slot[a].reserved = true;
        if (b >= 0) g_ctx.slot[b].reserved = true;
    }
    ~SlotReservation()
    {
        if (a >= 0) g_ctx.slot[a].reserved = false;
        if (b >= 0) g_ctx.slot[b].reserved = false;
    }
};
int ensure_slot_stream(MsmSlot& S)
{
    if (!S.stream) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    return BBGPU_OK;
}

// The pieces of the point range [off, off + n) of an entry: one per table segment it touches (one in all without tables or inside one segment).
struct PointPiece {
    const SrsEntry::TabSeg* seg; // null: no tables
    size_t off_in_seg, first, len; // first: offset inside the call's range (and its scalars)
};
int split_pieces(const SrsEntry& e, size_t off, size_t n, PointPiece* out, int cap)
{
    if (!e.has_tab()) {
        out[0] = PointPiece{ nullptr, off, 0, n };
        return 1;
    }
    if (n == 0) { // nothing to add up: any segment will do
        out[0] = PointPiece{ &e.segs[0], 0, 0, 0 };
        return 1;
    }
    int cnt = 0;
    for (const auto& sg : e.segs) {
        const size_t lo = std::max(off, sg.first), hi = std::min(off + n, sg.first + sg.n);
        if (lo >= hi) continue;
        if (cnt == cap) return -1;
        out[cnt++] = PointPiece{ &sg, lo - sg.first, lo - off, hi - lo };
    }
    return cnt;
}
constexpr int MAX_POINT_PIECES = MSM_MAX_PIECES;

bool others_pending(const MsmSlot* a, const MsmSlot* b = nullptr)
{
    for (int k = 0; k < Context::NSLOT; k++)
        if (&g_ctx.slot[k] != a && &g_ctx.slot[k] != b && g_ctx.slot[k].pending) return true;
    return false;
}
// collects whatever slot t (and its helper) still has in flight and forgets it: error paths
void drain_ticket(int t)
{
    MsmSlot& S = g_ctx.slot[t];
    host::Xyzz dump[MSM_MAX_JOBS];
    if (S.helper >= 0) {
        MsmSlot& H = g_ctx.slot[S.helper];
        if (H.pending) (void)msm_finish_batch(H, dump, nullptr);
        H.is_helper = false;
        S.helper = -1;
    }
    if (S.pending) (void)msm_finish_batch(S, dump, nullptr);
}
// Issues `jobs` MSMs (one scalar vector each) over points [off, off + n) of entry e, windows [wb, we), on slot t; `st` = the caller's stream or the
// slot's own.  Inside one table segment (every SRS up to 2^20 points) that is one pass through the kernels.  A range that spans several segments
// is issued as one PIECE per segment, dealt alternately to slot t and -- when one is free -- a HELPER slot with its own stream and workspace, so
// that the digit / sort front of piece k + 1 runs beside the accumulation of piece k exactly as two consecutive MSMs do (DESIGN_HISTORY 5); the helper
// is ordered behind the producer of the scalars by an event when the caller gave a stream.  The ticket stays slot t: finish_ticket() adds up
// both slots' pieces.
int issue_ticket(int t, const SrsEntry& e, size_t off, const uint64_t* const* d_scalars_v, int jobs, size_t n, int wb, int we, hipStream_t st)
{
    MsmSlot& S = g_ctx.slot[t];
    S.helper = -1;
    S.append = false;
    if (!windows_resident(e, wb, we)) return BBGPU_ERR_STATE;
    PointPiece pc[MAX_POINT_PIECES];
    const int np = split_pieces(e, off, n, pc, MAX_POINT_PIECES);
    if (np < 0) {
        set_error("MSM of %zu points spans more than %d table segments", n, MAX_POINT_PIECES);
        return BBGPU_ERR_SIZE;
    }
    if (np == 1) {
        S.throughput = others_pending(&S);
        const uint32_t* tab = pc[0].seg ? pc[0].seg->d_tab + pc[0].off_in_seg * 16 : nullptr;
        return msm_issue_batch(S, e.d_srs + off * 16, tab, pc[0].seg ? pc[0].seg->n : e.n, e.tab_c, d_scalars_v, jobs, n, wb, we, st, g_ctx.timing);
    }
    // several pieces: a helper slot for every other one, if any slot is free
    int h = -1;
    {
        int order[Context::NSLOT], cnt = 0;
        if (t < 2) order[cnt++] = t ^ 1; // the pair the two-deep pipeline of large MSMs uses
        for (int k = Context::NSLOT - 1; k >= 2; --k) order[cnt++] = k; // from the top: the low ones are what the next tickets take
        for (int k = 0; k < cnt && h < 0; k++)
            if (order[k] != t && !g_ctx.slot[order[k]].pending && !g_ctx.slot[order[k]].reserved) h = order[k];
    }
    MsmSlot* H = h >= 0 ? &g_ctx.slot[h] : nullptr;
    if (H) {
        if (int rc = ensure_slot_stream(*H)) return rc;
        // `st` carries the producer of the scalars -- a caller's kernel, or the asynchronous upload host_to_device() queued on the slot's OWN stream
        // (bbgpu_msm_g1 / _batch): the helper's stream starts behind what is enqueued there now, whichever stream that is
        if (!g_ctx.helper_dep[h]) CHK(hipEventCreateWithFlags(&g_ctx.helper_dep[h], hipEventDisableTiming));
        CHK(hipEventRecord(g_ctx.helper_dep[h], st));
        CHK(hipStreamWaitEvent(H->stream, g_ctx.helper_dep[h], 0));
        H->helper = -1;
    }
    int issued[2] = { 0, 0 };
    int rc = BBGPU_OK;
    // every slot's workspace is sized ONCE, for the largest piece it will carry: (seg / 2, 2 seg + 7) is pieces of seg / 2, seg and seg / 2 + 7 points, and a
    // workspace grown for the second piece would be freed (dev_free waits for the device) under the first one still queued on the stream
    {
        size_t largest[2] = { 0, 0 };
        for (int k = 0; k < np; k++) largest[(H && (k & 1)) ? 1 : 0] = std::max(largest[(H && (k & 1)) ? 1 : 0], pc[k].len);
        for (int side = 0; side < 2 && rc == BBGPU_OK; side++)
            if (largest[side]) rc = (side ? *H : S).ws.ensure(MsmWorkspace::bytes_needed(largest[side], e.tab_c, (we - wb) * jobs, jobs)); // pieces run against window tables: one bucket set per job
        if (rc != BBGPU_OK) return rc;
    }
    for (int k = 0; k < np && rc == BBGPU_OK; k++) {
        const int side = (H && (k & 1)) ? 1 : 0;
        MsmSlot& T = side ? *H : S;
        const uint64_t* sv[MSM_MAX_JOBS];
        for (int j = 0; j < jobs; j++) sv[j] = d_scalars_v[j] + pc[k].first * 4;
        T.append = issued[side] > 0;
        T.throughput = true; // pieces share the chip with each other
        rc = msm_issue_batch(T, e.d_srs + (off + pc[k].first) * 16, pc[k].seg->d_tab + pc[k].off_in_seg * 16, pc[k].seg->n, e.tab_c, sv, jobs, pc[k].len, wb, we,
                             side ? H->stream : st, g_ctx.timing);
        if (rc == BBGPU_OK) issued[side]++;
    }
    if (H && issued[1] > 0) {
        H->is_helper = true;
        S.helper = h;
    }
    if (rc != BBGPU_OK) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        drain_ticket(t);
        memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
}
int issue_on_entry(int t, const SrsEntry& e, size_t off, const uint64_t* d_scalars, size_t n, int wb, int we, hipStream_t st)
{
    return issue_ticket(t, e, off, &d_scalars, 1, n, wb, we, st);
}
// waits for ticket t and adds up its pieces: one point per job
int finish_ticket(int t, host::Xyzz* results, MsmTiming* timing)
{
    MsmSlot& S = g_ctx.slot[t];
    const uint32_t jobs = S.jobs;
    const int h = S.helper;
    S.helper = -1;
    int rc = msm_finish_batch(S, results, timing);
    if (h >= 0) {
        MsmSlot& H = g_ctx.slot[h];
        host::Xyzz more[MSM_MAX_JOBS];
        const int rc2 = msm_finish_batch(H, more, nullptr);
        H.is_helper = false;
        if (rc == BBGPU_OK) rc = rc2;
        if (rc == BBGPU_OK)
            for (uint32_t j = 0; j < jobs; j++) results[j] = host::g1_add(results[j], more[j]);
    }
    return rc;
}

// does the host range [points, points + n) still hold what entry e was uploaded from (rows off .. off + n)?
// 16 rows per check: the first, the last, and 14 evenly spaced ones whose PHASE moves on with every check of this entry, so that a buffer
// rewritten only in the middle (a slice the fixed sample never touched) is caught within a few calls instead of never; the residual window
// (a partial in-place rewrite is served stale until a sampled row falls into it) is stated in include/bbgpu.h.
bool contents_match(SrsEntry& e, size_t off, const uint64_t* points, size_t n)
{
    if (e.row_hash.size() != e.n) return false;
    auto same = [&](size_t i) { return hash_row(points + i * 16) == e.row_hash[off + i]; };
    if (n <= 16) {
        for (size_t i = 0; i < n; i++)
            if (!same(i)) return false;
        return true;
    }
    if (!same(0) || !same(n - 1)) return false;
    const size_t stride = n / 14, phase = (size_t)((e.check_phase++ * 0x9e3779b97f4a7c15ULL) % stride);
    for (size_t k = 0; k < 14; k++)
        if (!same(phase + k * stride)) return false;
    return true;
}

// EXACT mode: every row of the range against the fingerprints taken at upload, spread over the staging pool's threads (64 B read per point:
// 2^16 points ~0.05 ms, 2^20 ~1 ms on the boxes' hosts -- which is why the callers run it AFTER they have issued the call's kernels, beside them)
bool contents_match_full(const SrsEntry& e, size_t off, const uint64_t* points, size_t n)
{
    if (e.row_hash.size() != e.n) return false;
    struct Job { const uint64_t* p; const uint64_t* h; std::atomic<int> bad; } job{ points, e.row_hash.data() + off, { 0 } };
    g_copy_pool.for_range(n, (size_t)1 << 14, [](void* c, size_t lo, size_t hi) {
        Job* j = static_cast<Job*>(c);
        uint64_t diff = 0;
        for (size_t i = lo; i < hi; i++) diff |= hash_row(j->p + i * 16) ^ j->h[i];
        if (diff) j->bad.store(1);
    }, &job);
    return job.bad.load() == 0;
}

// The same check in the BACKGROUND (the staging pool's helper threads alone) for a call whose scalars do not cross the link through that pool (uploads above
// host_stage_max go to hipMemcpyAsync as they are): started before the upload, joined when the call's kernels have been issued -- the hash of a 2^20-point table
// (64 MiB of host memory) then runs beside the 32 MiB upload and the launches instead of after them.
struct FullCheckJob {
    const uint64_t* p = nullptr;
    const uint64_t* h = nullptr;
    std::atomic<int> bad{ 0 };
    bool posted = false;
};
void full_check_post(FullCheckJob& job, const SrsEntry& e, size_t off, const uint64_t* points, size_t n)
{
    job.p = points;
    job.h = e.row_hash.data() + off;
    job.bad.store(e.row_hash.size() != e.n ? 1 : 0);
    job.posted = true;
    if (job.bad.load()) return;
    g_copy_pool.post_range(n, [](void* c, size_t lo, size_t hi) {
        FullCheckJob* j = static_cast<FullCheckJob*>(c);
        uint64_t diff = 0;
        for (size_t i = lo; i < hi; i++) diff |= hash_row(j->p + i * 16) ^ j->h[i];
        if (diff) j->bad.store(1);
    }, &job);
}
bool full_check_join(FullCheckJob& job) // true: contents match
{
    g_copy_pool.join();
    job.posted = false;
    return job.bad.load() == 0;
}

// table lookup by host address, VALIDATED by content: returns entry index and point offset, or -1.  An auto-registered table whose
// address range matches but whose contents do not (the caller freed the table and another landed there, or refilled the buffer) is
// evicted; an explicitly registered one is merely not served (its handle stays valid for the device-pointer entries; in-place mutation
// of a registered table requires bbgpu_srs_release, see bbgpu.h).  Newest entries first.
// deferred_full (optional): set when the entry is in EXACT mode and the caller takes over the full check (contents_match_full, run beside the
// kernels it has issued; a mismatch there = srs_mark_stale + redo); without it the full check runs here.
int find_srs(const uint64_t* points, size_t n, size_t* offset, bool* deferred_full = nullptr)
{
    if (deferred_full) *deferred_full = false;
    for (size_t k = g_ctx.srs.size(); k-- > 0;) {
        SrsEntry& e = g_ctx.srs[k];
        if (!e.live || !e.host_ptr || e.stale_for_host) continue;
        const uint8_t* b = (const uint8_t*)e.host_ptr;
        const uint8_t* p = (const uint8_t*)points;
        if (p < b || p >= b + e.n * 128) continue;
        const size_t d = (size_t)(p - b);
        if (d % 128) continue;
        if (d / 128 + n > e.n) continue;
        if (!contents_match(e, d / 128, points, n) || (e.validate_full && !deferred_full && !contents_match_full(e, d / 128, points, n))) {
            if (trace_srs()) fprintf(stderr, "bbgpu srs: contents of %p (+%zu rows, n=%zu) differ from the resident copy\n", (const void*)e.host_ptr, d / 128, n);
            if (e.auto_registered) free_entry(e);
            else if (e.validate_full) e.stale_for_host = true;
            continue;
        }
        if (deferred_full) *deferred_full = e.validate_full;
        e.last_use = ++g_ctx.use_clock;
        *offset = d / 128;
        return (int)k;
    }
    return -1;
}

// the deferred full check of an exact-mode entry failed: never serve this copy to a host-pointer call again
void srs_mark_stale(int idx)
{
    SrsEntry& e = g_ctx.srs[idx];
    if (trace_srs()) fprintf(stderr, "bbgpu srs: full check: contents of %p differ from the resident copy\n", (const void*)e.host_ptr);
    if (e.auto_registered) free_entry(e);
    else e.stale_for_host = true;
}

int log2_exact(size_t n)
{
    if (n == 0 || (n & (n - 1))) return -1;
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// plain: `points` is an n-entry table of plain affine points (64 bytes apart) instead of the 2n-entry endomorphism table -- the argument of
// the reference's pippenger_low_memory / pippenger_precomputed (scalar_multiplication.cpp:142-262, :478-574).  Such a table is used once and
// forgotten (no address-keyed cache: these are test / bench entries of the reference, not the prover's).
int msm_host_ptrs_once(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t out[12], bool plain, bool* stale);
int msm_host_ptrs(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t out[12], bool plain = false)
{
    // EXACT cache mode: the call runs against the resident copy while the host re-hashes every row of the caller's table; if they differ the
    // copy is dropped and the call runs once more, now uploading the table as it is (the reference reads the caller's points on every call,
    // scalar_multiplication.cpp:604-617)
    bool stale = false;
    int rc = msm_host_ptrs_once(scalars, points, n, out, plain, &stale);
    if (stale) rc = msm_host_ptrs_once(scalars, points, n, out, plain, &stale);
    return rc;
}
int msm_host_ptrs_once(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t out[12], bool plain, bool* stale)
{
    *stale = false;
    if (n == 0) {
        host::g1_to_normalised(host::g1_infinity(), out);
        return BBGPU_OK;
    }
    if (!scalars || !points) {
        set_error("null scalars/points");
        return BBGPU_ERR_ARG;
    }
    size_t off = 0;
    bool full_check = false;
    int idx = plain ? -1 : find_srs(points, n, &off, &full_check);
    read_host_env();
    if (idx < 0 && n <= (size_t)g_ctx.host_msm_max) { // the verifier's ~20 freshly built points: no allocation, no launch
        host::g1_to_normalised(host::msm_small(scalars, points, n, plain ? 8 : 16), out);
        return BBGPU_OK;
    }
    {
        int rc = ensure_init();
        if (rc) return rc;
    }
    // whatever slots are free (a caller -- or another thread -- may hold asynchronous tickets on any of them): two give the pipeline, one works
    int sl[2];
    const int ns = free_slots(sl, 2);
    if (ns == 0) {
        set_error("all %d MSM slots are in flight: call bbgpu_msm_g1_wait first", Context::NSLOT);
        return BBGPU_ERR_STATE;
    }
    for (int k = 0; k < ns; k++)
        if (int rc = ensure_slot_stream(g_ctx.slot[sl[k]])) return rc;
    SlotReservation reserve(sl, ns);
    // A table that was never registered and is too small to be an SRS (the verifier's ~20 freshly built points,
    // verifier.cpp:359-363) is used once and forgotten: caching it by address would both leak device memory per call and
    // serve stale points when the caller's vector is freed and its address reused.  Larger unknown tables are taken to be a
    // long-lived SRS and registered on first sight.
    // exact cache mode on a table large enough that the scalars bypass the staging pool: the full check starts NOW, on the pool's helper threads, and is joined
    // once the kernels are issued (every way out of this function joins it first: the job reads the caller's table and the entry's fingerprints)
    FullCheckJob bg;
    struct BgGuard { FullCheckJob& j; ~BgGuard() { if (j.posted) (void)full_check_join(j); } } bg_guard{ bg };
    if (full_check && n * 32 > g_ctx.host_stage_max) full_check_post(bg, g_ctx.srs[idx], off, points, n);
    SrsEntry transient{};
    const bool is_transient = idx < 0 && (plain || n < AUTO_REGISTER_MIN_POINTS);
    const bool tr = trace_srs();
    auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double q0 = tr ? now_ms() : 0;
    // a small table that is used once (the verifier's freshly built points, test tables): uploaded into a buffer the library keeps, on the stream of the
    // slot that runs the call's one range -- no allocation, no free, no wait (round 5: those were ~0.05 of the 0.39 ms such a call took)
    const bool small_once = is_transient && n < AUTO_REGISTER_MIN_POINTS;
    if (small_once) {
        const size_t stride = plain ? 64 : 128;
        int rc = grow(&g_ctx.d_small_tab, &g_ctx.small_tab_cap, AUTO_REGISTER_MIN_POINTS * (128 + 64));
        if (rc) return rc;
        uint32_t* d_raw = (uint32_t*)g_ctx.d_small_tab;
        uint32_t* d = d_raw + AUTO_REGISTER_MIN_POINTS * 32;
        if ((rc = srs_upload_into(points, n, d_raw, d, g_ctx.slot[sl[0]].stream, stride)) != BBGPU_OK) return rc;
        transient.host_ptr = points;
        transient.n = n;
        transient.d_srs = d;
        transient.live = true;
        off = 0;
    } else
    if (idx < 0) {
        uint32_t* d = nullptr;
        int rc = srs_upload(points, n, &d, g_ctx.stream, plain ? 64 : 128);
        if (rc) return rc;
        if (is_transient) {
            transient.host_ptr = points;
            transient.n = n;
            transient.d_srs = d;
            transient.live = true;
        } else {
            idx = add_srs(points, n, d, true);
            if (idx < 0) return idx;
        }
        off = 0;
    }
    const SrsEntry& e = is_transient ? transient : g_ctx.srs[idx];
    // The call is cut into point RANGES that go through the free slots like the jobs of a batch: the scalars of range k+1 cross the link while
    // the kernels of range k run, and the partial sums (group elements: the sum over a range of points is a plain term of the whole sum) are
    // added on the host.  Above 2^20 points the ranges are the table segments the call touches (each at most 2^20 points with its own window
    // tables); inside one segment a call of 2^19 points and more is cut in two -- one 2^20-point call: 32 MiB of scalars = 0.6 ms on the link
    // before the first kernel, against 0.22 ms for the first of two ranges (bench.py `boundary`).  BBGPU_HOST_MSM_SPLIT=1 keeps one range per segment.
    static const size_t split_env = [] { const char* v = getenv("BBGPU_HOST_MSM_SPLIT"); return v ? (size_t)std::max(1, atoi(v)) : (size_t)0; }();
    struct Range { size_t o, len; };
    std::vector<Range> ranges;
    {
        PointPiece pc[MAX_POINT_PIECES];
        const int np = split_pieces(e, off, n, pc, MAX_POINT_PIECES);
        if (np < 0) {
            if (is_transient && !small_once) (void)dev_free(transient.d_srs);
            set_error("MSM of %zu points spans more than %d table segments", n, MAX_POINT_PIECES);
            return BBGPU_ERR_SIZE;
        }
        if (np > 1) {
            for (int k = 0; k < np; k++) ranges.push_back(Range{ pc[k].first, pc[k].len });
        } else {
            // Measured on MI355X (tools/boundary_ab.py, 2^20 points): one range 2.05-2.08 ms, two 1.77 ms, four 2.17-2.20 ms -- every range pays its
            // own sort and bucket-reduction tail (~0.3 ms of launches that only partly hide), so two it is, the first one the smaller: its
            // upload is the part nothing hides, and the second range's upload (0.6 ms x its share) still fits under the first one's kernels.
            const size_t parts = ns < 2 ? 1 : (split_env ? split_env : (n >= ((size_t)1 << 19) ? 2 : 1));
            static const size_t first_pct = [] { const char* v = getenv("BBGPU_HOST_MSM_FIRST_PCT"); return v ? (size_t)std::min(50, std::max(5, atoi(v))) : (size_t)0; }(); // tuning knob; measured 25 / 30 / 34 / 37 / 42 %: 1.84 / 1.80 / 1.83 / 1.775 / 1.79 ms
            const size_t base = parts == 2 ? (((first_pct ? n * first_pct / 100 : n * 3 / 8)) & ~(size_t)7) : n / parts;
            for (size_t k = 0; k < parts; k++) {
                const size_t o = k * base, len = (k + 1 == parts) ? n - o : base;
                if (len) ranges.push_back(Range{ o, len });
            }
        }
    }
    size_t max_len = 0;
    for (const auto& r : ranges) max_len = std::max(max_len, r.len);
    const double q1 = tr ? now_ms() : 0;
    uint64_t** stage[2] = { &g_ctx.d_stage, &g_ctx.d_stage2 };
    size_t* cap[2] = { &g_ctx.stage_cap, &g_ctx.stage2_cap };
    host::Xyzz res = host::g1_infinity();
    int rc = BBGPU_OK;
    size_t issued = 0, finished = 0;
    auto finish = [&](size_t k) -> int {
        host::Xyzz part;
        int r = finish_ticket(sl[k % (size_t)ns], &part, k + 1 == ranges.size() ? &g_ctx.last : nullptr);
        finished = k + 1;
        if (r == BBGPU_OK) res = host::g1_add(res, part);
        return r;
    };
    for (size_t k = 0; k < ranges.size() && rc == BBGPU_OK; k++) {
        const size_t w = k % (size_t)ns;
        if (k >= (size_t)ns) rc = finish(k - (size_t)ns); // frees this range's slot and staging buffer
        MsmSlot& S = g_ctx.slot[sl[w]];
        if (rc == BBGPU_OK) rc = grow(stage[w], cap[w], max_len * 32);
        if (rc == BBGPU_OK) rc = host_to_device(*stage[w], scalars + ranges[k].o * 4, ranges[k].len * 32, S.stream);
        if (rc == BBGPU_OK) rc = issue_on_entry(sl[w], e, off + ranges[k].o, *stage[w], ranges[k].len, 0, entry_windows(e, ranges[k].len), S.stream);
        if (rc == BBGPU_OK) issued = k + 1;
    }
    const double q2 = tr ? now_ms() : 0;
    // exact mode: every row of the caller's table against the fingerprints of the resident copy, on the host while the kernels issued above run
    if (rc == BBGPU_OK && full_check && !(bg.posted ? full_check_join(bg) : contents_match_full(g_ctx.srs[idx], off, points, n))) {
        for (int k = 0; k < ns; k++) drain_ticket(sl[k]);
        srs_mark_stale(idx);
        *stale = true;
        return BBGPU_OK;
    }
    while (rc == BBGPU_OK && finished < issued) rc = finish(finished);
    if (rc != BBGPU_OK) { // drain whatever is still in flight so that the slots are usable again (the error text is the first failure's)
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        for (int k = 0; k < ns; k++) drain_ticket(sl[k]);
        memcpy(g_err, keep, sizeof(keep));
    }
    const double q3 = tr ? now_ms() : 0;
    if (is_transient && !small_once) (void)dev_free(transient.d_srs); // the finishes have waited for the kernels
    if (rc) return rc;
    host::g1_to_normalised(res, out);
    if (tr) fprintf(stderr, "bbgpu msm n=%zu: table %.3f, upload + issue %.3f, wait + host sums %.3f, free + normalise %.3f ms\n", n, q1 - q0, q2 - q1, q3 - q2, now_ms() - q3);
    return BBGPU_OK;
}

} // namespace
} // namespace bbgpu

using namespace bbgpu;

#pragma GCC visibility push(default)
extern "C" {

const char* bbgpu_version(void) { return "bbgpu 0.1 (gfx950)"; }
const char* bbgpu_last_error(void) { return g_err; }

int bbgpu_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int bbgpu_init(int device)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g_ctx.ready && g_ctx.device == device) return BBGPU_OK;
    if (g_ctx.ready) {
        set_error("already bound to device %d (one process per GPU)", g_ctx.device);
        return BBGPU_ERR_STATE;
    }
    g_ctx.device = device;
    return ensure_init();
}

void bbgpu_shutdown(void)
{
    // lock order: the prover's mutex, then the library's (the prover calls the entry points above while it holds its own)
    std::lock_guard<std::mutex> lkp(plonk_mutex());
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_ctx.ready) return;
    plonk_release_all_locked();
    // nothing may still be reading the pinned staging buffers or a slot's workspace when they are freed: collect what is in flight
    // (an MSM a caller never waited for, a copy queued before an error return) and drain every stream first
    for (int k = 0; k < Context::NSLOT; k++)
        if (g_ctx.slot[k].pending && !g_ctx.slot[k].is_helper) drain_ticket(k);
    for (auto& ev : g_ctx.helper_dep) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
    }
    (void)hipStreamSynchronize(g_ctx.stream);
    for (auto& sl : g_ctx.slot)
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    (void)hipDeviceSynchronize();
    host_stage_release();
    g_ctx.poly_scratch.release();
    if (g_ctx.d_poly_tmp) (void)dev_free(g_ctx.d_poly_tmp);
    g_ctx.d_poly_tmp = nullptr;
    g_ctx.poly_tmp_cap = 0;
    for (auto& e : g_ctx.srs) {
        if (e.live && e.d_srs) (void)dev_free(e.d_srs);
        if (e.live)
            for (auto& sg : e.segs)
                if (sg.d_tab_alloc) (void)dev_free(sg.d_tab_alloc);
    }
    g_ctx.srs.clear();
    for (auto& sl : g_ctx.slot) sl.release();
    if (g_ctx.d_stage) (void)dev_free(g_ctx.d_stage);
    if (g_ctx.d_stage2) (void)dev_free(g_ctx.d_stage2);
    g_ctx.d_stage2 = nullptr;
    g_ctx.stage2_cap = 0;
    if (g_ctx.d_scratch) (void)dev_free(g_ctx.d_scratch);
    g_ctx.d_stage = g_ctx.d_scratch = nullptr;
    g_ctx.stage_cap = g_ctx.scratch_cap = 0;
    if (g_ctx.d_small_tab) (void)dev_free(g_ctx.d_small_tab);
    g_ctx.d_small_tab = nullptr;
    g_ctx.small_tab_cap = 0;
    ntt_release_tables();
    if (g_ctx.shared_done) (void)hipEventDestroy(g_ctx.shared_done);
    g_ctx.shared_done = nullptr;
    g_ctx.shared_used = false;
    (void)hipStreamDestroy(g_ctx.stream);
    g_ctx.stream = nullptr;
    g_ctx.ready = false;
}

int bbgpu_memory_stats(bbgpu_memory_info* out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return BBGPU_ERR_ARG;
    memset(out, 0, sizeof(*out));
    for (const auto& e : g_ctx.srs) {
        if (!e.live) continue;
        const uint64_t pts = (uint64_t)e.n * 64, tab = e.bytes > pts ? e.bytes - pts : 0;
        out->srs_points_bytes += pts;
        out->srs_table_bytes += tab;
        if (e.auto_registered) out->srs_auto_bytes += e.bytes;
    }
    out->srs_cache_cap_bytes = g_ctx.srs_cache_cap;
    size_t cap = 0;
    int sets = 0;
    out->ntt_table_bytes = ntt_table_bytes(&cap, &sets);
    out->ntt_table_cap_bytes = cap;
    out->ntt_table_sets = (uint64_t)sets;
    for (const auto& sl : g_ctx.slot) {
        out->msm_workspace_bytes += sl.ws.cap;
        if (sl.ws.h_out) out->pinned_host_bytes += (uint64_t)MSM_HOUT_GROUPS * 64 * 128;
    }
    out->staging_bytes = g_ctx.stage_cap + g_ctx.stage2_cap + g_ctx.scratch_cap + g_ctx.poly_tmp_cap + g_ctx.poly_scratch.cap + g_ctx.small_tab_cap;
    for (int k = 0; k < Context::HOST_RING; k++)
        if (g_ctx.h_stage[k]) out->pinned_host_bytes += Context::HOST_CHUNK;
    return BBGPU_OK;
}

int bbgpu_fault_inject(const char* spec)
{
    if (fault_set(spec)) {
        set_error("fault spec '%s' not understood (alloc:k | h2d:k | d2h:k | launch:k)", spec ? spec : "");
        return BBGPU_ERR_ARG;
    }
    return BBGPU_OK;
}
int bbgpu_fault_stats(bbgpu_fault_info* out)
{
    if (!out) return BBGPU_ERR_ARG;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    FaultState& F = fault();
    memset(out, 0, sizeof(*out));
    out->alloc_calls = F.calls[F_ALLOC].load();
    out->h2d_calls = F.calls[F_H2D].load();
    out->d2h_calls = F.calls[F_D2H].load();
    out->launch_checks = F.calls[F_LAUNCH].load();
    out->armed = F.armed_kind.load() >= 0 ? 1 : 0;
    out->fired = F.fired.load();
    out->absorbed = F.absorbed.load();
    {
        std::lock_guard<std::mutex> lf(F.mu);
        out->live_allocations = F.live.size();
        out->live_bytes = F.live_bytes;
    }
    for (const auto& sl : g_ctx.slot)
        if (sl.pending) out->slots_pending++;
    return BBGPU_OK;
}

void bbgpu_set_timing(int enabled)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_ctx.timing = enabled < 0 || enabled > 2 ? 1 : enabled;
}
int bbgpu_last_timing(float* ms_out, int max_entries)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int k = g_ctx.last.count < max_entries ? g_ctx.last.count : max_entries;
    for (int i = 0; i < k; i++) ms_out[i] = g_ctx.last.ms[i];
    return k;
}

/* ---- NTT ---- */
int bbgpu_ntt_device(uint64_t* d_coeffs, size_t n, int kind, const uint64_t* constant, void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    const int lg = log2_exact(n);
    if (lg < 1) {
        set_error("NTT size %zu is not a power of two >= 2", n);
        return BBGPU_ERR_SIZE;
    }
    if (kind < 0 || kind > BBGPU_COSET_FFT_WITH_CONSTANT || !d_coeffs) {
        set_error("bad NTT kind / null buffer");
        return BBGPU_ERR_ARG;
    }
    rc = grow(&g_ctx.d_scratch, &g_ctx.scratch_cap, n * 32);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream; /* NULL = the legacy default stream, as bbgpu.h says */
    if ((rc = shared_begin(st)) != BBGPU_OK) return rc;
    rc = ntt_device(d_coeffs, g_ctx.d_scratch, lg, kind, constant, st);
    if (rc == BBGPU_OK) rc = shared_end(st);
    if (rc == BBGPU_ERR_SIZE) set_error("NTT size 2^%d unsupported (max 2^28)", lg);
    if (rc == BBGPU_ERR_HIP) {
        const hipError_t he = hipGetLastError(); // hipSuccess: the failing call has already described itself (launch_check / dev_malloc)
        if (he != hipSuccess) set_error("NTT launch failed: %s", hipGetErrorString(he));
    }
    return rc;
}

int bbgpu_ntt_device_batch(uint64_t* d_coeffs, size_t n, size_t stride_elems, int batch, int kind, const uint64_t* constant, void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    const int lg = log2_exact(n);
    if (lg < 1) {
        set_error("NTT size %zu is not a power of two >= 2", n);
        return BBGPU_ERR_SIZE;
    }
    if (kind < 0 || kind > BBGPU_COSET_FFT_WITH_CONSTANT || !d_coeffs || batch < 1 || batch > 64 || stride_elems < n) {
        set_error("bad NTT kind / null buffer / batch / stride");
        return BBGPU_ERR_ARG;
    }
    rc = grow(&g_ctx.d_scratch, &g_ctx.scratch_cap, (size_t)batch * n * 32);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream; /* NULL = the legacy default stream, as bbgpu.h says */
    if ((rc = shared_begin(st)) != BBGPU_OK) return rc;
    rc = ntt_device_batch(d_coeffs, stride_elems, batch, g_ctx.d_scratch, lg, kind, constant, st);
    if (rc == BBGPU_OK) rc = shared_end(st);
    if (rc == BBGPU_ERR_SIZE) set_error("NTT size 2^%d unsupported (max 2^28)", lg);
    if (rc == BBGPU_ERR_HIP) {
        const hipError_t he = hipGetLastError(); // hipSuccess: the failing call has already described itself (launch_check / dev_malloc)
        if (he != hipSuccess) set_error("NTT launch failed: %s", hipGetErrorString(he));
    }
    return rc;
}

int bbgpu_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!coeffs) return BBGPU_ERR_ARG;
    read_host_env();
    if (n <= (size_t)g_ctx.host_ntt_max && log2_exact(n) >= 1 && kind >= 0 && kind <= BBGPU_COSET_FFT_WITH_CONSTANT) {
        const bool has_const = (kind == BBGPU_FFT_WITH_CONSTANT || kind == BBGPU_IFFT_WITH_CONSTANT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
        if (has_const && !constant) return BBGPU_ERR_ARG;
        host::ntt_small(coeffs, log2_exact(n), kind, constant); // SURVEY 8b small sizes: no copy, no launch
        return BBGPU_OK;
    }
    int rc = ensure_init();
    if (rc) return rc;
    rc = grow(&g_ctx.d_stage, &g_ctx.stage_cap, n * 32);
    if (rc) return rc;
    if ((rc = host_to_device(g_ctx.d_stage, coeffs, n * 32, g_ctx.stream)) != BBGPU_OK) return rc;
    rc = bbgpu_ntt_device(g_ctx.d_stage, n, kind, constant, g_ctx.stream);
    if (rc) return rc;
    // up to here `coeffs` is untouched; a failure while the result is copied back may leave it half overwritten -- LOST only if a byte of it was written
    bool touched = false;
    rc = device_to_host_sync(coeffs, g_ctx.d_stage, n * 32, g_ctx.stream, &touched);
    return rc == BBGPU_OK ? BBGPU_OK : (touched ? BBGPU_ERR_LOST : rc);
}

/* ---- resident polynomial helpers ---- */
struct SharedGuard { // records "this stream is done with the shared workspaces" on every way out of a helper
    hipStream_t st;
    ~SharedGuard() { (void)shared_end(st); }
};
#define POLY_ENTER(ptr_ok)                                                                                              \
    std::lock_guard<std::recursive_mutex> lk(g_mu);                                                                     \
    {                                                                                                                   \
        int rc_ = ensure_init();                                                                                        \
        if (rc_) return rc_;                                                                                            \
        if (!(ptr_ok)) {                                                                                                \
            set_error("null device pointer");                                                                           \
            return BBGPU_ERR_ARG;                                                                                       \
        }                                                                                                               \
    }                                                                                                                   \
    hipStream_t st = (hipStream_t)hip_stream; /* NULL = the legacy default stream, as bbgpu.h says */                                               \
    if (int rcb_ = shared_begin(st)) return rcb_;                                                                       \
    SharedGuard shared_guard_{ st }

static host::Fr load_fr(const uint64_t z[4])
{
    host::Fr r;
    memcpy(r.d, z, 32);
    return r;
}

int bbgpu_fr_evaluate_device(const uint64_t* d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4], void* hip_stream)
{
    POLY_ENTER((d_coeffs || n == 0) && z && out);
    host::Fr r;
    int rc = poly::evaluate(d_coeffs, n, load_fr(z), &r, g_ctx.poly_scratch, st);
    if (rc) return rc;
    memcpy(out, r.d, 32);
    return BBGPU_OK;
}

int bbgpu_fr_batch_invert_device(uint64_t* d_values, size_t n, void* hip_stream)
{
    POLY_ENTER(d_values || n == 0);
    int rc = grow(&g_ctx.d_poly_tmp, &g_ctx.poly_tmp_cap, n * 32);
    if (rc) return rc;
    return poly::batch_invert(d_values, g_ctx.d_poly_tmp, n, g_ctx.poly_scratch, st);
}

int bbgpu_fr_product_scan_device(const uint64_t* d_in, uint64_t* d_out, size_t n, int reverse, int inclusive, void* hip_stream)
{
    POLY_ENTER((d_in && d_out) || n == 0);
    return poly::product_scan(d_in, d_out, n, reverse != 0, inclusive != 0, g_ctx.poly_scratch, st, nullptr);
}

int bbgpu_fr_mul_device(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, void* hip_stream)
{
    POLY_ENTER((d_out && d_a && d_b) || n == 0);
    return poly::mul_pointwise(d_out, d_a, d_b, n, st);
}

int bbgpu_kate_opening_device(const uint64_t* d_src, uint64_t* d_dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4], void* hip_stream)
{
    POLY_ENTER(((d_src && d_dest) || n == 0) && z);
    const host::Fr zz = load_fr(z);
    if (f_of_z) {
        host::Fr f;
        int rc = poly::evaluate(d_src, n, zz, &f, g_ctx.poly_scratch, st);
        if (rc) return rc;
        memcpy(f_of_z, f.d, 32);
    }
    const uint64_t* src = d_src;
    if (d_dest == d_src) { // the scan's last phase reads its input while writing: work from a copy
        int rc = grow(&g_ctx.d_poly_tmp, &g_ctx.poly_tmp_cap, n * 32);
        if (rc) return rc;
        CHK(hipMemcpyAsync(g_ctx.d_poly_tmp, d_src, n * 32, hipMemcpyDeviceToDevice, st));
        src = g_ctx.d_poly_tmp;
    }
    return poly::horner_suffix(src, d_dest, n, zz, false, g_ctx.poly_scratch, st, nullptr);
}

int bbgpu_lagrange_l1_fft_device(uint64_t* d_l_1, size_t n_src, size_t n_target, void* hip_stream)
{
    POLY_ENTER(d_l_1);
    const int ls = log2_exact(n_src), lt = log2_exact(n_target);
    if (ls < 1 || lt < ls || lt > 28) {
        set_error("lagrange_l1_fft: domains must be powers of two, target >= source");
        return BBGPU_ERR_SIZE;
    }
    int rc = grow(&g_ctx.d_poly_tmp, &g_ctx.poly_tmp_cap, n_target * 32);
    if (rc) return rc;
    return poly::lagrange_l1_fft(d_l_1, g_ctx.d_poly_tmp, ls, lt, g_ctx.poly_scratch, st);
}

int bbgpu_divide_by_pseudo_vanishing_device(uint64_t* d_coeffs, size_t n_src, size_t n_target, void* hip_stream)
{
    POLY_ENTER(d_coeffs);
    const int ls = log2_exact(n_src), lt = log2_exact(n_target);
    if (ls < 1 || lt < ls || lt > 28) {
        set_error("divide_by_pseudo_vanishing: domains must be powers of two, target >= source");
        return BBGPU_ERR_SIZE;
    }
    return poly::divide_by_pseudo_vanishing(d_coeffs, ls, lt, st);
}

int bbgpu_permutation_lagrange_base_device(uint64_t* d_out, const uint32_t* d_mapping, size_t n, void* hip_stream)
{
    POLY_ENTER(d_out && d_mapping);
    const int lg = log2_exact(n);
    if (lg < 1 || lg > 28) return BBGPU_ERR_SIZE;
    int rc = grow(&g_ctx.d_poly_tmp, &g_ctx.poly_tmp_cap, n * 32);
    if (rc) return rc;
    rc = poly::powers(g_ctx.d_poly_tmp, n, host::fr_root_of_unity(lg), host::fr_one(), st);
    if (rc) return rc;
    return poly::sigma_from_mapping(d_out, d_mapping, g_ctx.d_poly_tmp, n, st);
}

/* ---- the same helpers on host buffers (what the C++ shim forwards the reference's co-resident TU functions to) ---- */
static int stage_in(const uint64_t* host, size_t n)
{
    int rc = grow(&g_ctx.d_stage, &g_ctx.stage_cap, n * 32);
    if (rc) return rc;
    return host_to_device(g_ctx.d_stage, host, n * 32, g_ctx.stream);
}
// in_place: the destination is also the call's input -- a failure after any byte of it was written is BBGPU_ERR_LOST (nothing left to fall back on)
static int stage_out(uint64_t* host, const uint64_t* dev, size_t n, bool in_place = false)
{
    bool touched = false;
    const int rc = device_to_host_sync(host, dev, n * 32, g_ctx.stream, &touched);
    return (rc != BBGPU_OK && in_place && touched) ? BBGPU_ERR_LOST : rc;
}

int bbgpu_fr_evaluate(const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if ((!coeffs && n) || !z || !out) return BBGPU_ERR_ARG;
    if ((rc = stage_in(coeffs, n)) != BBGPU_OK) return rc;
    return bbgpu_fr_evaluate_device(g_ctx.d_stage, n, z, out, g_ctx.stream);
}

int bbgpu_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (((!src || !dest) && n) || !z) return BBGPU_ERR_ARG;
    if ((rc = stage_in(src, n)) != BBGPU_OK) return rc;
    if ((rc = grow(&g_ctx.d_stage2, &g_ctx.stage2_cap, n * 32)) != BBGPU_OK) return rc;
    if ((rc = bbgpu_kate_opening_device(g_ctx.d_stage, g_ctx.d_stage2, n, z, f_of_z, g_ctx.stream)) != BBGPU_OK) return rc;
    // the reference calls it in place (polynomial.cpp:327 passes coefficients, coefficients): then a half-written dest is a half-destroyed src
    return stage_out(dest, g_ctx.d_stage2, n, dest == src);
}

int bbgpu_lagrange_l1_fft(uint64_t* l_1, size_t n_src, size_t n_target)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (!l_1) return BBGPU_ERR_ARG;
    if ((rc = grow(&g_ctx.d_stage, &g_ctx.stage_cap, n_target * 32)) != BBGPU_OK) return rc;
    if ((rc = bbgpu_lagrange_l1_fft_device(g_ctx.d_stage, n_src, n_target, g_ctx.stream)) != BBGPU_OK) return rc;
    return stage_out(l_1, g_ctx.d_stage, n_target);
}

int bbgpu_divide_by_pseudo_vanishing(uint64_t* coeffs, size_t n_src, size_t n_target)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (!coeffs) return BBGPU_ERR_ARG;
    if ((rc = stage_in(coeffs, n_target)) != BBGPU_OK) return rc;
    if ((rc = bbgpu_divide_by_pseudo_vanishing_device(g_ctx.d_stage, n_src, n_target, g_ctx.stream)) != BBGPU_OK) return rc;
    return stage_out(coeffs, g_ctx.d_stage, n_target, true); // in place, as in bbgpu_ntt
}

// polynomial_arithmetic::get_lagrange_evaluations (polynomial_arithmetic.cpp:594-626): {Z_H*(z), L_1(z), L_{n-1}(z)}; host arithmetic
int bbgpu_lagrange_evaluations(const uint64_t z[4], size_t n, uint64_t out[12])
{
    const int lg = log2_exact(n);
    if (!z || !out || lg < 1) return BBGPU_ERR_ARG;
    const host::Fr zc = load_fr(z), one = host::fr_one(), root = host::fr_root_of_unity(lg), root_inv = host::fr_inv(root);
    host::Fr zp = zc;
    for (int i = 0; i < lg; i++) zp = host::fr_sqr(zp);
    const host::Fr numerator = host::fr_sub(zp, one);
    const host::Fr d0 = host::fr_inv(host::fr_sub(zc, root_inv)), d1 = host::fr_inv(host::fr_sub(zc, one));
    const host::Fr d2 = host::fr_inv(host::fr_sub(host::fr_mul(host::fr_mul(zc, root), root), one));
    const host::Fr scaled = host::fr_mul(numerator, host::fr_inv(host::fr_from_u64((uint64_t)n)));
    const host::Fr v = host::fr_mul(numerator, d0), l1 = host::fr_mul(scaled, d1), ln = host::fr_mul(scaled, d2);
    memcpy(out, v.d, 32);
    memcpy(out + 4, l1.d, 32);
    memcpy(out + 8, ln.d, 32);
    return BBGPU_OK;
}

// scalar_multiplication::generate_pippenger_point_table (scalar_multiplication.cpp:131-140): table[2i] = P_i, table[2i+1] = (beta x_i, -y_i),
// filled from the back so that `points` may alias `table`.  Host arithmetic (once per SRS).
int bbgpu_generate_point_table(const uint64_t* points, uint64_t* table, size_t n)
{
    if ((!points || !table) && n) return BBGPU_ERR_ARG;
    const host::Fq beta = { { 0x71930c11d782e155ULL, 0xa6bb947cffbe3323ULL, 0xaa303344d4741444ULL, 0x2c3b3f0d26594943ULL } }; // fq.hpp:53-56
    const host::Fq zero = { { 0, 0, 0, 0 } };
    for (size_t i = n; i-- > 0;) {
        host::Fq x, y;
        memcpy(x.d, points + i * 8, 32);
        memcpy(y.d, points + i * 8 + 4, 32);
        const host::Fq bx = host::fq_mul(x, beta), ny = host::fq_sub(zero, y);
        uint64_t* e = table + i * 16;
        memcpy(e + 8, bx.d, 32);
        memcpy(e + 12, ny.d, 32);
        memcpy(e, x.d, 32);
        memcpy(e + 4, y.d, 32);
    }
    return BBGPU_OK;
}

/* ---- host fallbacks: what shim/bb_shim.cpp computes with after a GPU entry has FAILED (host_fallback.hpp).  No HIP call, no lock,
 * no shared state: re-entrant.  Never reached from the GPU entries above. ---- */
int bbgpu_host_msm_g1(const uint64_t* scalars, const uint64_t* points, size_t n, int plain_table, uint64_t out[12])
{
    if (!out || (n && (!scalars || !points))) {
        set_error("null scalars / points / out");
        return BBGPU_ERR_ARG;
    }
    host::g1_to_normalised(host::msm_pippenger(scalars, points, n, plain_table ? 8 : 16), out);
    return BBGPU_OK;
}
int bbgpu_host_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant)
{
    const int lg = log2_exact(n);
    if (lg < 1 || lg > 28) {
        set_error("NTT size %zu is not a power of two in [2, 2^28]", n);
        return BBGPU_ERR_SIZE;
    }
    const bool has_const = (kind == BBGPU_FFT_WITH_CONSTANT || kind == BBGPU_IFFT_WITH_CONSTANT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    if (!coeffs || kind < 0 || kind > BBGPU_COSET_FFT_WITH_CONSTANT || (has_const && !constant)) {
        set_error("bad NTT kind / null buffer");
        return BBGPU_ERR_ARG;
    }
    host::ntt_radix2(coeffs, lg, kind, constant);
    return BBGPU_OK;
}
int bbgpu_host_fr_evaluate(const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4])
{
    if ((!coeffs && n) || !z || !out) return BBGPU_ERR_ARG;
    const host::Fr r = host::poly_evaluate(coeffs, n, load_fr(z));
    memcpy(out, r.d, 32);
    return BBGPU_OK;
}
int bbgpu_host_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_of_z[4])
{
    if (((!src || !dest) && n) || !z) return BBGPU_ERR_ARG;
    const host::Fr f = host::kate_opening(src, dest, n, load_fr(z));
    if (f_of_z) memcpy(f_of_z, f.d, 32);
    return BBGPU_OK;
}
int bbgpu_host_lagrange_l1_fft(uint64_t* l_1, size_t n_src, size_t n_target)
{
    const int ls = log2_exact(n_src), lt = log2_exact(n_target);
    if (!l_1) return BBGPU_ERR_ARG;
    if (ls < 1 || lt < ls || lt > 28) return BBGPU_ERR_SIZE;
    host::lagrange_l1_fft(l_1, ls, lt);
    return BBGPU_OK;
}
int bbgpu_host_divide_by_pseudo_vanishing(uint64_t* coeffs, size_t n_src, size_t n_target)
{
    const int ls = log2_exact(n_src), lt = log2_exact(n_target);
    if (!coeffs) return BBGPU_ERR_ARG;
    if (ls < 1 || lt < ls || lt > 28) return BBGPU_ERR_SIZE;
    host::divide_by_pseudo_vanishing(coeffs, ls, lt);
    return BBGPU_OK;
}

/* ---- SRS ---- */
int bbgpu_srs_register(const uint64_t* points_endo_table, size_t n)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (!points_endo_table || n == 0) return BBGPU_ERR_ARG;
    size_t off;
    int idx = find_srs(points_endo_table, n, &off);
    if (idx >= 0 && off == 0) {
        g_ctx.srs[idx].auto_registered = false; // the caller now holds the handle: never evicted behind its back
        g_ctx.srs[idx].handle_exposed = true;
        return idx;
    }
    uint32_t* d = nullptr;
    rc = srs_upload(points_endo_table, n, &d, g_ctx.stream);
    if (rc) return rc;
    return add_srs(points_endo_table, n, d, false);
}

int bbgpu_srs_generate_range(const uint64_t* x_mont, size_t first, size_t n, uint64_t* host_endo_table_out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (!x_mont || n == 0 || first > ((size_t)1 << 31) || n > ((size_t)1 << 31)) return BBGPU_ERR_ARG;
    uint32_t* d = nullptr;
    rc = srs_generate(x_mont, first, n, &d, host_endo_table_out, g_ctx.stream);
    if (rc) return rc;
    return add_srs(host_endo_table_out, n, d, false);
}
int bbgpu_srs_generate(const uint64_t* x_mont, size_t n, uint64_t* host_endo_table_out)
{
    return bbgpu_srs_generate_range(x_mont, 0, n, host_endo_table_out);
}

// io.hpp:36-182 restated for the G1 part: 28-byte manifest of seven big-endian uint32 (fields 5 = num_g1_points), then points as
// x, y, each four 64-bit limbs least-significant limb first, every limb big-endian, NOT in Montgomery form; file point k is
// x^(k+1) G and monomials[0] is the generator (read_transcript :159-181).  Fills the complete 2n-entry endomorphism table of
// generate_pippenger_point_table (scalar_multiplication.cpp:131-140): entry 2i = P_i, entry 2i+1 = (beta x_i, -y_i).  Host only.
int bbgpu_transcript_read_g1(const char* path, size_t degree, uint64_t* points_endo_table_out)
{
    if (!path || !points_endo_table_out || degree == 0) return BBGPU_ERR_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open transcript %s", path);
        return BBGPU_ERR_ARG;
    }
    unsigned char man[28];
    if (fread(man, 1, 28, f) != 28) {
        fclose(f);
        set_error("transcript %s: short manifest", path);
        return BBGPU_ERR_SIZE;
    }
    auto be32 = [&](int i) { return ((uint32_t)man[4 * i] << 24) | ((uint32_t)man[4 * i + 1] << 16) | ((uint32_t)man[4 * i + 2] << 8) | man[4 * i + 3]; };
    const uint32_t num_g1 = be32(4);
    if ((size_t)num_g1 + 1 < degree) {
        fclose(f);
        set_error("transcript %s holds %u G1 points, %zu needed", path, num_g1, degree - 1);
        return BBGPU_ERR_SIZE;
    }
    const host::Fq rsq = { { 0xF32CFC5B538AFA89ULL, 0xB5E71911D44501FBULL, 0x47AB1EFF0A417FF6ULL, 0x06D89F71CAB8351FULL } }; // 2^512 mod q (fq.hpp:48-51)
    const host::Fq beta = { { 0x71930c11d782e155ULL, 0xa6bb947cffbe3323ULL, 0xaa303344d4741444ULL, 0x2c3b3f0d26594943ULL } }; // fq.hpp:53-56 (Montgomery)
    const host::Fq zero = { { 0, 0, 0, 0 } };
    auto put = [&](size_t i, const host::Fq& x, const host::Fq& y) {
        uint64_t* e = points_endo_table_out + i * 16;
        memcpy(e, x.d, 32);
        memcpy(e + 4, y.d, 32);
        const host::Fq bx = host::fq_mul(x, beta), ny = host::fq_sub(zero, y);
        memcpy(e + 8, bx.d, 32);
        memcpy(e + 12, ny.d, 32);
    };
    host::Fq two = host::fq_add(host::FQ_ONE, host::FQ_ONE);
    put(0, host::FQ_ONE, two); // g1::affine_one = (1, 2) (g1.hpp:14-16)
    std::vector<unsigned char> buf(64 * 4096);
    size_t done = 1;
    while (done < degree) {
        const size_t chunk = std::min<size_t>(4096, degree - done);
        if (fread(buf.data(), 64, chunk, f) != chunk) {
            fclose(f);
            set_error("transcript %s: short read", path);
            return BBGPU_ERR_SIZE;
        }
        for (size_t k = 0; k < chunk; k++) {
            host::Fq c[2];
            for (int xy = 0; xy < 2; xy++)
                for (int l = 0; l < 4; l++) {
                    uint64_t v = 0;
                    for (int b = 0; b < 8; b++) v = (v << 8) | buf[k * 64 + xy * 32 + l * 8 + b];
                    c[xy].d[l] = v;
                }
            put(done + k, host::fq_mul(c[0], rsq), host::fq_mul(c[1], rsq));
        }
        done += chunk;
    }
    fclose(f);
    return BBGPU_OK;
}

// io.hpp:36-135,159-181 restated for WRITING: the file read_transcript(monomials, g2_x, degree, path) accepts for this SRS.
// 28-byte manifest (seven big-endian uint32: transcript_number 0, total_transcripts 1, total_g1_points, total_g2_points 2, num_g1_points,
// num_g2_points 2, start_from 0), the degree - 1 points x G .. x^(degree-1) G (entries 2 .. 2 (degree - 1) of the endo table; entry 0, the
// generator, is implicit in the format), then G2 and x G2, then the 64-byte checksum slot the reference never verifies (zeros).
// Coordinates leave Montgomery form; four 64-bit limbs least-significant first, each limb big-endian.  Host only.
int bbgpu_transcript_write(const char* path, const uint64_t* points_endo_table, size_t degree, const uint64_t x_mont[4])
{
    if (!path || !points_endo_table || degree < 2 || !x_mont || degree - 1 > 0xffffffffu) return BBGPU_ERR_ARG;
    host::G2Affine xg2;
    if (!host::g2_scalar_mul_affine(host::G2_ONE, load_fr(x_mont), &xg2)) {
        set_error("transcript secret is zero");
        return BBGPU_ERR_ARG;
    }
    FILE* f = fopen(path, "wb");
    if (!f) {
        set_error("cannot create transcript %s", path);
        return BBGPU_ERR_ARG;
    }
    const host::Fq one_raw = { { 1, 0, 0, 0 } };
    auto put_fq = [&](unsigned char* dst, const uint64_t* mont) {
        host::Fq v;
        memcpy(v.d, mont, 32);
        v = host::fq_mul(v, one_raw); // out of Montgomery form
        for (int l = 0; l < 4; l++)
            for (int b = 0; b < 8; b++) dst[l * 8 + b] = (unsigned char)(v.d[l] >> (8 * (7 - b)));
    };
    const uint32_t man[7] = { 0, 1, (uint32_t)(degree - 1), 2, (uint32_t)(degree - 1), 2, 0 };
    unsigned char mb[28];
    for (int i = 0; i < 7; i++)
        for (int b = 0; b < 4; b++) mb[4 * i + b] = (unsigned char)(man[i] >> (8 * (3 - b)));
    bool ok = fwrite(mb, 1, 28, f) == 28;
    std::vector<unsigned char> buf(64 * 4096);
    for (size_t done = 1; ok && done < degree;) {
        const size_t chunk = std::min<size_t>(4096, degree - done);
        for (size_t k = 0; k < chunk; k++) {
            const uint64_t* e = points_endo_table + (done + k) * 16;
            put_fq(&buf[k * 64], e);
            put_fq(&buf[k * 64 + 32], e + 4);
        }
        ok = fwrite(buf.data(), 64, chunk, f) == chunk;
        done += chunk;
    }
    unsigned char g2b[2 * 128 + 64];
    memset(g2b, 0, sizeof(g2b));
    const host::G2Affine pts[2] = { host::G2_ONE, xg2 };
    for (int i = 0; i < 2; i++) { // g2::affine_element = {x.c0, x.c1, y.c0, y.c1} (io.hpp:100-135)
        put_fq(g2b + 128 * i, pts[i].x.c0.d);
        put_fq(g2b + 128 * i + 32, pts[i].x.c1.d);
        put_fq(g2b + 128 * i + 64, pts[i].y.c0.d);
        put_fq(g2b + 128 * i + 96, pts[i].y.c1.d);
    }
    ok = ok && fwrite(g2b, 1, sizeof(g2b), f) == sizeof(g2b);
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error("short write to transcript %s", path);
        return BBGPU_ERR_ARG;
    }
    return BBGPU_OK;
}

int bbgpu_srs_set_validate(int srs_handle, int full)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (srs_handle == -1) {
        g_ctx.srs_validate_full = full != 0;
        return BBGPU_OK;
    }
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) return BBGPU_ERR_ARG;
    g_ctx.srs[srs_handle].validate_full = full != 0;
    return BBGPU_OK;
}

int bbgpu_srs_release(int handle)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (handle < 0 || handle >= (int)g_ctx.srs.size() || !g_ctx.srs[handle].live) return BBGPU_ERR_ARG;
    free_entry(g_ctx.srs[handle]);
    return BBGPU_OK;
}

// resident tables right now: how many, how many of them registered on first sight (evictable), device bytes held by those
int bbgpu_srs_cache_stats(int* live_entries, int* auto_entries, uint64_t* auto_bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int live = 0, au = 0;
    uint64_t bytes = 0;
    for (const auto& e : g_ctx.srs)
        if (e.live) {
            live++;
            if (e.auto_registered) {
                au++;
                bytes += e.bytes;
            }
        }
    if (live_entries) *live_entries = live;
    if (auto_entries) *auto_entries = au;
    if (auto_bytes) *auto_bytes = bytes;
    return BBGPU_OK;
}

/* ---- MSM ---- */
int bbgpu_msm_num_windows(size_t n)
{
    return msm_num_windows(msm_choose_c(n ? n : 1));
}
int bbgpu_srs_num_windows(int srs_handle, size_t n)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) return BBGPU_ERR_ARG;
    return entry_windows(g_ctx.srs[srs_handle], n);
}
// Multi-GPU: rank `rank` of `world` will only ever be asked for its 1/world share of the (window, point) rows of tables registered
// from now on, so only the digit windows that share touches are built and kept: 15 x 64 MiB at n = 2^20 become ceil(15 / world) + 1 windows.
void bbgpu_set_table_share(int rank, int world)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (world < 1 || rank < 0 || rank >= world) { rank = 0; world = 1; }
    g_ctx.share_rank = rank;
    g_ctx.share_world = world;
}

// Multi-GPU, the other split: a rank holds n / world POINTS of a larger MSM as its own SRS (all digit windows of them) and its MSM over the matching
// scalars is its partial sum.  Tables registered from now on pick their window size as the whole MSM would.
void bbgpu_set_point_share(int world)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_ctx.point_world = world >= 1 && world <= 1024 ? world : 1;
}

void bbgpu_set_precompute(int enabled)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_ctx.precompute = enabled != 0;
}

int bbgpu_msm_g1(const uint64_t* scalars, const uint64_t* points_endo_table, size_t n, uint64_t out[12])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return msm_host_ptrs(scalars, points_endo_table, n, out); // binds the device itself unless the host answers (n = 0, tiny unknown tables)
}

int bbgpu_msm_g1_plain(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t out[12])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return msm_host_ptrs(scalars, points, n, out, true);
}

void bbgpu_set_host_thresholds(int msm_max_points, int ntt_max_elements)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_ctx.host_env_read = true;
    g_ctx.host_msm_max = msm_max_points < 0 ? 0 : msm_max_points;
    g_ctx.host_ntt_max = ntt_max_elements < 0 ? 0 : std::min(64, ntt_max_elements);
}

static int msm_g1_batch_once(bbgpu_msm_job* jobs, size_t num_jobs, bool* stale);
int bbgpu_msm_g1_batch(bbgpu_msm_job* jobs, size_t num_jobs)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bool stale = false; // exact cache mode: a resident table turned out to differ from the caller's memory -- dropped; the batch runs once more on a fresh upload
    int rc = msm_g1_batch_once(jobs, num_jobs, &stale);
    if (stale) rc = msm_g1_batch_once(jobs, num_jobs, &stale);
    return rc;
}
static int msm_g1_batch_once(bbgpu_msm_job* jobs, size_t num_jobs, bool* stale)
{
    *stale = false;
    int rc = BBGPU_OK;
    if (num_jobs == 0) return BBGPU_OK;
    if (!jobs) return BBGPU_ERR_ARG;
    for (size_t i = 1; i < num_jobs; i++) {
        if (jobs[i].num_elements != jobs[0].num_elements) {
            // scalar_multiplication.cpp:678-685: report and leave the outputs untouched
            set_error("batched_scalar_multiplications err: each scalar mul must be same size.");
            return BBGPU_ERR_ARG;
        }
    }
    const size_t n = jobs[0].num_elements;
    read_host_env();
    int sl[2];
    // one at a time: tiny jobs (answered on the host), fewer than two free slots, or jobs above one table segment (each is a pipeline of its own)
    bool one_by_one = n == 0 || n <= (size_t)g_ctx.host_msm_max || n > ((size_t)1 << 20) || free_slots(sl, 2) < 2;
    if (one_by_one) {
        for (size_t i = 0; i < num_jobs; i++) {
            rc = msm_host_ptrs(jobs[i].scalars, jobs[i].points, jobs[i].num_elements, jobs[i].output);
            if (rc) return rc;
        }
        return BBGPU_OK;
    }
    if ((rc = ensure_init()) != BBGPU_OK) return rc;
    for (int k = 0; k < 2; k++)
        if ((rc = ensure_slot_stream(g_ctx.slot[sl[k]])) != BBGPU_OK) return rc;
    SlotReservation reserve(sl, 2); // a job that straddles two table segments takes its helper elsewhere (or none)
    // Two-slot pipeline over the jobs of a prover round (3/1/3/2 MSMs, prover.cpp:65-122,650-658): job i+1's scalars
    // cross PCIe and its kernels are enqueued while job i's bucket-reduction tail and host finish run.
    uint64_t** stage[2] = { &g_ctx.d_stage, &g_ctx.d_stage2 };
    size_t* cap[2] = { &g_ctx.stage_cap, &g_ctx.stage2_cap };
    auto now_ms = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const bool tr = trace_srs();
    struct FullCheck { int idx; size_t off; const uint64_t* points; };
    std::vector<FullCheck> to_check;
    auto issue = [&](size_t i) -> int {
        const double q0 = tr ? now_ms() : 0;
        const int t = (int)(i & 1);
        MsmSlot& S = g_ctx.slot[sl[t]];
        if (!jobs[i].scalars || !jobs[i].points) {
            set_error("null scalars/points in job %zu", i);
            return BBGPU_ERR_ARG;
        }
        size_t off = 0;
        bool full_check = false;
        int idx = find_srs(jobs[i].points, n, &off, &full_check);
        if (idx < 0) {
            uint32_t* d = nullptr;
            int r = srs_upload(jobs[i].points, n, &d, g_ctx.stream);
            if (r) return r;
            idx = add_srs(jobs[i].points, n, d, true);
            if (idx < 0) return idx;
        }
        if (full_check) { // exact mode: checked in full once per distinct range of the batch, after the last job is issued (beside the kernels)
            bool known = false;
            for (const auto& c : to_check) known = known || (c.idx == idx && c.off == off);
            if (!known) to_check.push_back(FullCheck{ idx, off, jobs[i].points });
        }
        const double q1 = tr ? now_ms() : 0;
        int r = grow(stage[t], cap[t], n * 32);
        if (r) return r;
        if ((r = host_to_device(*stage[t], jobs[i].scalars, n * 32, S.stream)) != BBGPU_OK) return r;
        const double q2 = tr ? now_ms() : 0;
        r = issue_on_entry(sl[t], g_ctx.srs[idx], off, *stage[t], n, 0, entry_windows(g_ctx.srs[idx], n), S.stream);
        if (tr) fprintf(stderr, "bbgpu batch: job %zu srs %.3f, copy call %.3f, kernel launches %.3f ms\n", i, q1 - q0, q2 - q1, now_ms() - q2);
        return r;
    };
    auto finish = [&](size_t i) -> int {
        host::Xyzz res;
        int r = finish_ticket(sl[i & 1], &res, nullptr);
        if (r) return r;
        host::g1_to_normalised(res, jobs[i].output);
        return BBGPU_OK;
    };
    for (size_t i = 0; i < num_jobs && rc == BBGPU_OK; i++) {
        const double t0 = tr ? now_ms() : 0;
        rc = issue(i);
        const double t1 = tr ? now_ms() : 0;
        if (rc == BBGPU_OK && i >= 1) rc = finish(i - 1);
        if (tr) fprintf(stderr, "bbgpu batch: job %zu issue %.3f ms, finish(prev) %.3f ms\n", i, t1 - t0, now_ms() - t1);
    }
    const double t2 = tr ? now_ms() : 0;
    if (rc == BBGPU_OK) {
        bool bad = false;
        for (const auto& c : to_check)
            if (g_ctx.srs[c.idx].live && !contents_match_full(g_ctx.srs[c.idx], c.off, c.points, n)) {
                srs_mark_stale(c.idx);
                bad = true;
            }
        if (bad) { // outputs already written came from the stale copy: the rerun overwrites every one of them
            for (int k = 0; k < 2; k++) drain_ticket(sl[k]);
            *stale = true;
            return BBGPU_OK;
        }
    }
    if (rc == BBGPU_OK) rc = finish(num_jobs - 1);
    if (tr) fprintf(stderr, "bbgpu batch: last finish %.3f ms\n", now_ms() - t2);
    if (rc != BBGPU_OK) { // nothing of this call stays in flight (the error text is the first failure's)
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        for (int k = 0; k < 2; k++) drain_ticket(sl[k]);
        memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
}

int bbgpu_msm_g1_device(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int window_begin, int window_end,
                        uint64_t out[12], void* hip_stream)
{
    int ticket = bbgpu_msm_g1_device_async(srs_handle, offset, d_scalars, n, window_begin, window_end, hip_stream);
    if (ticket < 0) return ticket;
    return bbgpu_msm_g1_wait(ticket, out);
}

int bbgpu_msm_g1_device_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int window_begin, int window_end,
                              void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) {
        set_error("unknown SRS handle %d", srs_handle);
        return BBGPU_ERR_ARG;
    }
    const SrsEntry& e = g_ctx.srs[srs_handle];
    if (offset + n > e.n || (!d_scalars && n)) {
        set_error("MSM range [%zu, %zu) outside the registered table of %zu points", offset, offset + n, e.n);
        return BBGPU_ERR_ARG;
    }
    // slots 0 and 1 alternate (the two-deep pipeline of consecutive large MSMs: measured 1.50 ms/step against 1.72 when four
    // streams rotate -- more streams than hardware queues delay the next MSM's sort behind the previous one's tail);
    // slots 2 and 3 only take the overflow when both are busy (a prover round's three side-by-side commitments)
    const int t = pick_slot();
    if (t < 0) return BBGPU_ERR_STATE;
    MsmSlot& S = g_ctx.slot[t];
    if (!S.stream) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : S.stream;
    rc = issue_on_entry(t, e, offset, d_scalars, n, window_begin, window_end, st);
    if (rc == BBGPU_ERR_ARG) set_error("bad window range [%d, %d)", window_begin, window_end);
    if (rc) return rc;
    if (t < 2) g_ctx.next_slot = t ^ 1;
    return t;
}

int bbgpu_srs_has_window_tables(int srs_handle)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) return BBGPU_ERR_ARG;
    return g_ctx.srs[srs_handle].has_tab() ? 1 : 0;
}

int bbgpu_msm_g1_device_rows_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, uint64_t row_begin, uint64_t row_end,
                                   void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) {
        set_error("unknown SRS handle %d", srs_handle);
        return BBGPU_ERR_ARG;
    }
    const SrsEntry& e = g_ctx.srs[srs_handle];
    if (offset + n > e.n || !d_scalars || n == 0) {
        set_error("MSM range [%zu, %zu) outside the registered table of %zu points", offset, offset + n, e.n);
        return BBGPU_ERR_ARG;
    }
    if (e.segs.size() != 1) {
        set_error(e.has_tab() ? "row-range shares need ONE table segment: this SRS of %zu points keeps %zu (split it by point range instead)"
                              : "row-range shares need the pre-shifted window tables (one shared bucket set): this table has none", e.n, e.segs.size());
        return BBGPU_ERR_STATE;
    }
    const int t = pick_slot();
    if (t < 0) return BBGPU_ERR_STATE;
    MsmSlot& S = g_ctx.slot[t];
    if (!S.stream) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : S.stream;
    if (n && row_end > row_begin && !windows_resident(e, (int)(row_begin / n), (int)((row_end + n - 1) / n))) return BBGPU_ERR_STATE;
    S.helper = -1;
    S.append = false;
    S.throughput = others_pending(&S);
    rc = msm_issue_rows(S, e.d_srs + offset * 16, e.segs[0].d_tab + offset * 16, e.n, e.tab_c, d_scalars, n, row_begin, row_end, st, g_ctx.timing);
    if (rc == BBGPU_ERR_ARG) set_error("bad row range [%llu, %llu) of %d x %zu", (unsigned long long)row_begin, (unsigned long long)row_end, e.tab_W, n);
    if (rc) return rc;
    if (t < 2) g_ctx.next_slot = t ^ 1;
    return t;
}

int bbgpu_msm_g1_device_buckets_async(int srs_handle, size_t offset, const uint64_t* d_scalars, size_t n, int share, int share_count, void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) {
        set_error("unknown SRS handle %d", srs_handle);
        return BBGPU_ERR_ARG;
    }
    const SrsEntry& e = g_ctx.srs[srs_handle];
    if (offset + n > e.n || !d_scalars || n == 0) {
        set_error("MSM range [%zu, %zu) outside the registered table of %zu points", offset, offset + n, e.n);
        return BBGPU_ERR_ARG;
    }
    if (e.segs.size() != 1) {
        set_error(e.has_tab() ? "bucket-range shares need ONE table segment: this SRS of %zu points keeps %zu (split it by point range instead)"
                              : "bucket-range shares need the pre-shifted window tables (one shared bucket set): this table has none", e.n, e.segs.size());
        return BBGPU_ERR_STATE;
    }
    const int t = pick_slot();
    if (t < 0) return BBGPU_ERR_STATE;
    MsmSlot& S = g_ctx.slot[t];
    if (!S.stream) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : S.stream;
    if (!windows_resident(e, 0, e.tab_W)) return BBGPU_ERR_STATE; // every share reads every window's table
    if (share < 0 || share_count < 1 || share >= share_count) {
        set_error("bad bucket share %d of %d", share, share_count);
        return BBGPU_ERR_ARG;
    }
    S.helper = -1;
    S.append = false;
    S.throughput = others_pending(&S);
    rc = msm_issue_buckets(S, e.d_srs + offset * 16, e.segs[0].d_tab + offset * 16, e.n, e.tab_c, d_scalars, n, (uint32_t)share, (uint32_t)share_count, st, g_ctx.timing);
    if (rc == BBGPU_ERR_ARG) set_error("bad bucket share %d of %d (at most one share per row of the bucket matrix)", share, share_count);
    if (rc) return rc;
    if (t < 2) g_ctx.next_slot = t ^ 1;
    return t;
}

int bbgpu_msm_g1_device_batch_async(int srs_handle, size_t offset, const uint64_t* const* d_scalars, int jobs, size_t n, void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ensure_init();
    if (rc) return rc;
    if (srs_handle < 0 || srs_handle >= (int)g_ctx.srs.size() || !g_ctx.srs[srs_handle].live) {
        set_error("unknown SRS handle %d", srs_handle);
        return BBGPU_ERR_ARG;
    }
    const SrsEntry& e = g_ctx.srs[srs_handle];
    if (!d_scalars || jobs < 1 || offset + n > e.n) {
        set_error("bad batch: jobs %d, range [%zu, %zu) of %zu points", jobs, offset, offset + n, e.n);
        return BBGPU_ERR_ARG;
    }
    if ((!e.has_tab() && jobs > 1) || jobs > MSM_MAX_JOBS) {
        set_error("batched MSM: 1..%d jobs over an SRS registered with window tables (bbgpu_set_precompute, n >= 1024)", MSM_MAX_JOBS);
        return BBGPU_ERR_ARG;
    }
    const int t = pick_slot();
    if (t < 0) return BBGPU_ERR_STATE;
    MsmSlot& S = g_ctx.slot[t];
    if (!S.stream) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : S.stream;
    rc = issue_ticket(t, e, offset, d_scalars, jobs, n, 0, entry_windows(e, n), st);
    if (rc) return rc;
    if (t < 2) g_ctx.next_slot = t ^ 1;
    return t;
}

// The blocking part of a wait runs WITHOUT the library mutex: the events of the ticket's slot (and of its helper) are picked up under the lock,
// waited for outside it, and only the host finish -- which then finds them complete -- takes the lock again.  A thread that collects tickets
// therefore never stops another one from issuing (a rank of a multi-GPU split issues from one thread and collects / exchanges on another,
// barretenberg_amd/sharding.py; the reference calls pippenger() from an OpenMP region).  The caller's part of the contract is the usual one: a
// ticket is waited for once, by one thread.
static void wait_ticket_events_unlocked(int ticket)
{
    hipEvent_t ev[2] = { nullptr, nullptr };
    {
        std::lock_guard<std::recursive_mutex> lk(g_mu);
        if (ticket < 0 || ticket >= Context::NSLOT || !g_ctx.slot[ticket].pending || g_ctx.slot[ticket].is_helper) return; // the locked part reports it
        const MsmSlot& S = g_ctx.slot[ticket];
        ev[0] = S.done;
        if (S.helper >= 0) ev[1] = g_ctx.slot[S.helper].done;
    }
    for (hipEvent_t e : ev)
        if (e) (void)hipEventSynchronize(e); // errors surface in the locked finish, which synchronises again
}

int bbgpu_msm_g1_batch_wait(int ticket, uint64_t* out)
{
    wait_ticket_events_unlocked(ticket);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (ticket < 0 || ticket >= Context::NSLOT || !g_ctx.slot[ticket].pending || g_ctx.slot[ticket].is_helper || !out) {
        set_error("no MSM batch in flight for ticket %d", ticket);
        return BBGPU_ERR_ARG;
    }
    host::Xyzz res[MSM_MAX_JOBS];
    const uint32_t jobs = g_ctx.slot[ticket].jobs;
    int rc = finish_ticket(ticket, res, &g_ctx.last);
    if (rc) return rc;
    host::g1_batch_to_normalised(res, jobs, out); // one inversion for the whole batch
    return BBGPU_OK;
}

int bbgpu_msm_g1_wait(int ticket, uint64_t out[12])
{
    wait_ticket_events_unlocked(ticket);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (ticket < 0 || ticket >= Context::NSLOT || !g_ctx.slot[ticket].pending || g_ctx.slot[ticket].is_helper || g_ctx.slot[ticket].jobs != 1) {
        set_error("no MSM in flight for ticket %d", ticket);
        return BBGPU_ERR_ARG;
    }
    host::Xyzz res;
    int rc = finish_ticket(ticket, &res, &g_ctx.last);
    if (rc) return rc;
    host::g1_to_normalised(res, out);
    return BBGPU_OK;
}

int bbgpu_g1_sum(const uint64_t* points12, size_t count, uint64_t out[12])
{
    if (!out || (count && !points12)) return BBGPU_ERR_ARG;
    host::Xyzz acc = host::g1_infinity();
    for (size_t i = 0; i < count; i++) acc = host::g1_add(acc, host::g1_from_jacobian(points12 + 12 * i));
    host::g1_to_normalised(acc, out);
    return BBGPU_OK;
}

} // extern "C"
#pragma GCC visibility pop
