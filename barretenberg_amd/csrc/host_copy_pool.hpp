// host_copy_pool.hpp -- the CPU half of the staged host <-> device path (capi.hip host_to_device / device_to_host_sync): a copy of a
// few MiB between the caller's buffer and the library's pinned staging buffer, spread over a small pool of helper threads.
// One thread copies at 30-50 GB/s on the boxes' EPYC 9575F, the link takes 56 GB/s: with the caller's thread alone the staged path
// is bound by the copy (profiles/r03_shim_profile_2e16.json: 312 MB per 2^16-gate proof of the reference prover = ~9 ms of copying
// beside 5.6 ms on the link).  BBGPU_STAGE_THREADS helpers (default 3; 0 = the caller's thread only) take equal parts of every copy of
// 512 KiB or more.  Started on first use, joined by shutdown() / at process exit; no GPU calls on these threads.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace bbgpu {
namespace host {

class CopyPool {
public:
    typedef void (*RangeFn)(void* ctx, size_t lo, size_t hi);
    // memcpy(dst, src, bytes), possibly by several threads; returns when all of it is done
    void copy(void* dst, const void* src, size_t bytes)
    {
        join();
        const int helpers = bytes >= kMinParallel ? ensure_started() : 0;
        if (helpers == 0) {
            std::memcpy(dst, src, bytes);
            return;
        }
#ifdef BBGPU_COPY_POOL_TEST_DELAY // tests/cpp/test_host_sanitize.cpp: give freshly started helpers time to run before the job is posted
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
#endif
        const size_t parts = (size_t)helpers + 1;
        const size_t part = ((bytes / parts) + 4095) & ~(size_t)4095; // page-sized pieces
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = nullptr;
            dst_ = (char*)dst;
            src_ = (const char*)src;
            bytes_ = bytes;
            part_ = part;
            pending_ = helpers;
            ++generation_;
        }
        cv_.notify_all();
        run_part(0, (char*)dst, (const char*)src, bytes, part);
        // the helpers' parts are short (< 1 ms): spin, then yield
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins)
            if (spins > 2000) std::this_thread::yield();
    }
    // The same job on the HELPERS alone, in the background: post_range() returns at once, join() waits for it.  For work that should run beside
    // something the calling thread does meanwhile (the full content check of a large cached point table beside the upload of the call's scalars and
    // the launches, capi.hip).  Without helpers the job runs inside post_range().  copy() / for_range() join a posted job before they start theirs.
    void post_range(size_t items, RangeFn fn, void* ctx)
    {
        join();
        const int helpers = ensure_started();
        if (helpers == 0 || items == 0) {
            if (items) fn(ctx, 0, items);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = fn;
            ctx_ = ctx;
            bytes_ = items;
            part_ = (items + (size_t)helpers - 1) / (size_t)helpers;
            first_part_ = 1; // helper idx (1-based) takes piece idx - 1
            pending_ = helpers;
            posted_ = true;
            ++generation_;
        }
        cv_.notify_all();
    }
    void join()
    {
        if (!posted_) return;
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins)
            if (spins > 2000) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(mu_);
        fn_ = nullptr;
        first_part_ = 0;
        posted_ = false;
    }
    // fn(lo, hi) over disjoint pieces of [0, items), the caller's thread taking the first piece; returns when every piece is done.
    // ctx/fn are plain pointers (no allocation on this path).  Used for the full content check of a cached point table (capi.hip).
    void for_range(size_t items, size_t min_parallel, RangeFn fn, void* ctx)
    {
        join();
        const int helpers = items >= min_parallel ? ensure_started() : 0;
        if (helpers == 0) {
            fn(ctx, 0, items);
            return;
        }
        const size_t parts = (size_t)helpers + 1;
        const size_t part = (items + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = fn;
            ctx_ = ctx;
            bytes_ = items;
            part_ = part;
            pending_ = helpers;
            ++generation_;
        }
        cv_.notify_all();
        fn(ctx, 0, std::min(part, items));
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins)
            if (spins > 2000) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(mu_);
        fn_ = nullptr;
    }
    void shutdown()
    {
        join();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!started_) return;
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
        threads_.clear();
        started_ = false;
        stop_ = false;
    }
    ~CopyPool() { shutdown(); }

private:
    static constexpr size_t kMinParallel = (size_t)512 << 10;
    static void run_part(size_t idx, char* dst, const char* src, size_t bytes, size_t part)
    {
        const size_t lo = idx * part;
        if (lo >= bytes) return;
        std::memcpy(dst + lo, src + lo, std::min(part, bytes - lo));
    }
    int ensure_started()
    {
        if (started_) return (int)threads_.size();
        int n = 3;
        if (const char* e = std::getenv("BBGPU_STAGE_THREADS")) n = std::max(0, std::min(15, std::atoi(e)));
        std::lock_guard<std::mutex> lk(mu_);
        started_ = true;
        // a helper starts from the CURRENT generation: after a shutdown() / restart the counter is not zero, and a helper that compared it with
        // zero would "see" a job at once and copy between the pointers of the last job before the shutdown (a freed staging buffer)
        const unsigned long long g0 = generation_;
        for (int i = 0; i < n; i++) threads_.emplace_back([this, i, g0] { worker((size_t)i + 1, g0); });
        return n;
    }
    void worker(size_t idx, unsigned long long seen)
    {
        for (;;) {
            char* dst;
            const char* src;
            size_t bytes, part;
            RangeFn fn;
            void* ctx;
            size_t first;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                dst = dst_; src = src_; bytes = bytes_; part = part_; fn = fn_; ctx = ctx_; first = first_part_;
            }
            if (fn) {
                const size_t lo = (idx - first) * part;
                if (lo < bytes) fn(ctx, lo, std::min(lo + part, bytes));
            } else
            run_part(idx, dst, src, bytes, part);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    bool started_ = false, stop_ = false;
    unsigned long long generation_ = 0;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t bytes_ = 0, part_ = 0;
    RangeFn fn_ = nullptr; // null: the job is a memcpy of bytes_ bytes in part_-sized pieces; else fn_(ctx_, lo, hi) over [0, bytes_) items
    void* ctx_ = nullptr;
    size_t first_part_ = 0; // 1 while a posted (helpers-only) job is in flight: helper idx takes piece idx - 1
    bool posted_ = false;   // touched by the producing thread only
    std::atomic<int> pending_{ 0 };
};

} // namespace host
} // namespace bbgpu
