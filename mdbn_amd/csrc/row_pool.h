// RowPool: persistent CPU worker threads that gather table rows, out[r] = table[idx[r]] -- the host half of the row
// feeder (mdbn_feeder_*, mdbn_host_gather_rows; mdbn_capi.hip).  Plain C++ (no HIP): this header is also compiled alone
// under -fsanitize=thread by tests/test_tsan_row_pool.py.
#pragma once
#include <atomic>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__) || defined(__i386__)
#include <emmintrin.h>
#define MDBN_ROWPOOL_X86 1
#else
#define MDBN_ROWPOOL_X86 0
#endif

namespace mdbn_host {
struct RowPool {                                  // persistent worker threads: out[r] = table[idx[r]]
    // Workers SPIN for a job for up to 2 ms after the last one before they sleep on the condition variable: a feed delivers a
    // minibatch every ~170 us, and waking a sleeping thread costs 50+ us each on the GPU box's host (8.4 MB gathered in
    // 228 us with sleeping workers, 85 us with spinning ones).  An idle pool sleeps.
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work;
    const float* src = nullptr; int64_t ld_src = 0, n_rows = 0, cols = 0;
    const int64_t* idx = nullptr; int64_t n = 0;
    float* dst = nullptr; int64_t ld_dst = 0;
    std::atomic<int64_t> next{0};
    std::atomic<int> bad{0};
    std::atomic<uint64_t> generation{0};
    std::atomic<int> running{0}, sleepers{0};
    std::atomic<bool> stop{false};

    // MDBN_FEED_SPIN_US (environment, read once): how long an idle worker spins before it sleeps; 0 = sleep at once (no
    // core is kept busy between minibatches, at the price of a wake-up per job)
    static double spin_window()
    {
        static const double w = [] { const char* e = getenv("MDBN_FEED_SPIN_US"); return e ? std::max(0.0, atof(e)) * 1e-6 : 2e-3; }();
        return w;
    }
    static void relax()
    {
#if MDBN_ROWPOOL_X86
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // one row into the staging slot with NON-TEMPORAL stores: the slot is read next by the copy engine, not by a CPU, and a
    // plain memcpy first reads every destination line for ownership (3 bytes of traffic per byte copied instead of 2)
    static void copy_row(float* d, const float* s, size_t bytes)
    {
#if !MDBN_ROWPOOL_X86
        memcpy(d, s, bytes);
#else
        if ((reinterpret_cast<uintptr_t>(d) & 15u) != 0 || bytes < 256) { memcpy(d, s, bytes); return; }
        const __m128i* sp = reinterpret_cast<const __m128i*>(s);
        __m128i* dp = reinterpret_cast<__m128i*>(d);
        size_t q = bytes / 64;
        for (; q > 0; --q, sp += 4, dp += 4) {
            const __m128i a = _mm_loadu_si128(sp), b = _mm_loadu_si128(sp + 1), c = _mm_loadu_si128(sp + 2), e = _mm_loadu_si128(sp + 3);
            _mm_stream_si128(dp, a); _mm_stream_si128(dp + 1, b); _mm_stream_si128(dp + 2, c); _mm_stream_si128(dp + 3, e);
        }
        const size_t done = bytes / 64 * 64;
        if (done < bytes) memcpy(reinterpret_cast<char*>(d) + done, reinterpret_cast<const char*>(s) + done, bytes - done);
#endif
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    explicit RowPool(int threads)
    {
        for (int t = 0; t < threads; ++t) workers.emplace_back([this] { loop(); });
    }
    ~RowPool()
    {
        { std::lock_guard<std::mutex> l(m); stop.store(true); }
        cv_work.notify_all();
        for (auto& w : workers) w.join();
    }
    void rows()                                   // a few rows at a time: contiguous row copies, dynamic balance
    {
        constexpr int64_t CHUNK = 4;
        for (;;) {
            const int64_t r0 = next.fetch_add(CHUNK);
            if (r0 >= n) break;
            for (int64_t r = r0; r < std::min(n, r0 + CHUNK); ++r) {
                const int64_t s = idx ? idx[r] : r;
                if (s < 0 || s >= n_rows) { bad.store(1); continue; }
                copy_row(dst + r * ld_dst, src + s * ld_src, (size_t)cols * sizeof(float));
            }
        }
#if MDBN_ROWPOOL_X86
        _mm_sfence();                             // the streamed rows are globally visible before this thread checks in
#endif
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            const double give_up = now() + spin_window();
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == seen && !stop.load()) {
                relax();
                if ((++spins & 1023) == 0 && now() > give_up) {
                    std::unique_lock<std::mutex> l(m);
                    sleepers.fetch_add(1);
                    cv_work.wait(l, [&] { return stop.load() || generation.load(std::memory_order_acquire) != seen; });
                    sleepers.fetch_sub(1);
                }
            }
            if (stop.load()) return;
            seen = generation.load(std::memory_order_acquire);
            rows();
            running.fetch_sub(1, std::memory_order_release);
        }
    }
    // one gather at a time (the feeder's dispatcher, or mdbn_host_gather_rows's private pool)
    bool gather(const float* src_, int64_t n_rows_, int64_t cols_, int64_t ld_src_, const int64_t* idx_, int64_t n_,
                float* dst_, int64_t ld_dst_)
    {
        src = src_; n_rows = n_rows_; cols = cols_; ld_src = ld_src_; idx = idx_; n = n_; dst = dst_; ld_dst = ld_dst_;
        next.store(0); bad.store(0);
        running.store((int)workers.size());
        {
            std::lock_guard<std::mutex> l(m);     // a worker between its predicate and its wait holds this lock
            generation.fetch_add(1, std::memory_order_release);
        }
        if (sleepers.load() > 0) cv_work.notify_all();
        rows();                                   // the calling thread works too
        int spins = 0;
        while (running.load(std::memory_order_acquire) != 0) {      // every worker checks in: none still reads this job
            relax();
            if ((++spins & 4095) == 0) std::this_thread::yield();
        }
        return bad.load() == 0;
    }
};
}  // namespace mdbn_host
