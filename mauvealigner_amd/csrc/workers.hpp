// workers.hpp -- a small spin pool for the host loops between the kernels.
//
// The host side of one alignment contains short data-parallel loops (column assembly: 0.7 ms at bacterial scale)
// between single-threaded stages.  The helpers sleep on a condition variable; a stage that wants them arms the
// pool (one wake-up), they spin for jobs until it is disarmed, and go back to sleep.  parallel_for never depends
// on a helper having woken up: the calling thread takes chunks from the same counter, so with no helper awake it
// simply runs the whole range itself.  (Keeping the helpers spinning through a whole mauve_align call was measured
// to slow the single-threaded chaining stage by 20-40 %, so they are armed per stage; see default_threads.)
//
// Protocol (all sequentially consistent atomics): the caller publishes the job fields, opens the job and bumps
// seq_; a helper that sees a new seq_ enters (inflight_++), re-checks that the job is still open, takes chunks,
// and leaves (inflight_--).  The caller closes the job when every item is done and waits for inflight_ == 0
// before the fields may change again: either a late helper sees the job closed, or the caller sees it inside.
// Every loop run through the pool writes disjoint outputs, so results do not depend on the schedule.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

class SpinPool {
public:
    // threads = total including the caller; <= 1 means no helpers (every loop runs inline)
    explicit SpinPool(int threads)
    {
        for (int i = 1; i < threads; i++) th_.emplace_back([this] { helper_main(); });
    }
    ~SpinPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; armed_.store(false); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    SpinPool(const SpinPool &) = delete;
    SpinPool &operator=(const SpinPool &) = delete;

    static int default_threads()
    {
        // Opt-in.  On the bench host (16-CPU cgroup share of a 256-thread box) helpers cut the assembly from 0.8 to
        // 0.45 ms but the single-threaded stages of the same call ran 20 % slower whenever the process had more
        // runnable threads, a net loss; so the default is the calling thread alone.
        if (const char *e = getenv("MAUVE_HOST_THREADS")) { const int v = atoi(e); return v < 1 ? 1 : (v > 16 ? 16 : v); }
        return 1;
    }

    void arm()
    {
        if (th_.empty()) return;
        { std::lock_guard<std::mutex> lk(mu_); armed_.store(true); }
        cv_.notify_all();
    }
    void disarm() { armed_.store(false); }

    // f(begin, end) over [0, n) in chunks of `grain`
    template <class F> void parallel_for(int64_t n, int64_t grain, F &&f)
    {
        if (n <= 0) return;
        if (grain < 1) grain = 1;
        if (th_.empty() || !armed_.load() || n <= grain) { f((int64_t)0, n); return; }
        arg_ = &f;
        call_ = [](void *a, int64_t b, int64_t e) { (*static_cast<typename std::remove_reference<F>::type *>(a))(b, e); };
        end_ = n; grain_ = grain;
        next_.store(0); done_.store(0);
        open_.store(true);
        seq_.fetch_add(1);
        take_chunks();
        while (done_.load() < n) cpu_relax();
        open_.store(false);
        while (inflight_.load() != 0) cpu_relax();
    }

    // RAII: arm for the duration of one API call
    struct Armed {
        SpinPool *p;
        explicit Armed(SpinPool *pool) : p(pool) { if (p) p->arm(); }
        ~Armed() { if (p) p->disarm(); }
        Armed(const Armed &) = delete;
        Armed &operator=(const Armed &) = delete;
    };

private:
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void take_chunks()
    {
        for (;;) {
            const int64_t b = next_.fetch_add(grain_);
            if (b >= end_) break;
            const int64_t e = b + grain_ < end_ ? b + grain_ : end_;
            call_(arg_, b, e);
            done_.fetch_add(e - b);
        }
    }
    void helper_main()
    {
        uint64_t seen = seq_.load();
        for (;;) {
            if (!armed_.load()) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || armed_.load(); });
                if (stop_) return;
                continue;
            }
            const uint64_t s = seq_.load();
            if (s == seen) { cpu_relax(); continue; }
            seen = s;
            inflight_.fetch_add(1);
            if (open_.load()) take_chunks();
            inflight_.fetch_sub(1);
        }
    }

    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    std::atomic<bool> armed_{false};
    std::atomic<uint64_t> seq_{0};
    std::atomic<bool> open_{false};
    std::atomic<int> inflight_{0};
    std::atomic<int64_t> next_{0}, done_{0};
    int64_t end_ = 0, grain_ = 1;
    void (*call_)(void *, int64_t, int64_t) = nullptr;
    void *arg_ = nullptr;
};
