// ORACLE / TEST INFRASTRUCTURE: stand-in for oneTBB's parallel_for, so that the reference's own sources compile here without oneTBB.
// Serial by default (IPCREF_THREADS unset or 1): the loops run front to back on the calling thread, which is what every committed fixture
// was generated with.  IPCREF_THREADS=n (round 4, for the timed `cpu_reference` of bench.py): the same loop bodies -- written by the
// reference for concurrent execution under TBB (Energy.cpp:203-327, SelfCollisionHandler.cpp:71,427,578, Optimizer.cpp) -- are dealt in
// blocks to a pool of n std::threads (the caller is one of them); a parallel_for reached from inside a worker runs serially on that worker,
// like a nested TBB loop that finds no idle thread.  See mini_eigen.hpp.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
namespace tbb {
namespace detail_shim {
inline int env_threads()
{
    static const int n = [] {
        const char* e = std::getenv("IPCREF_THREADS");
        const int v = e ? std::atoi(e) : 1;
        return v > 1 ? v : 1;
    }();
    return n;
}
inline int& limit()
{ // tbb::global_control(max_allowed_parallelism, n) of main.cpp:821
    static int n = 1 << 30;
    return n;
}
inline bool& inside()
{
    static thread_local bool f = false;
    return f;
}
// Workers wait for the next loop by SPINNING on a generation counter for a few tens of microseconds before they go to sleep on a condition variable
// (the reference issues thousands of short loops per Newton iteration: waking 255 sleepers through a mutex for each of them took minutes on a
// 256-core box), and a loop only takes as many threads as it has blocks of GRAIN iterations.
class Pool {
    static constexpr long long GRAIN = 64;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned long long> gen_{ 0 };
    std::atomic<int> pending_{ 0 }, sleepers_{ 0 };
    std::atomic<bool> stop_{ false };
    const std::function<void(long long, long long)>* body_ = nullptr;
    std::atomic<long long> next_{ 0 };
    long long end_ = 0, chunk_ = 1;
    int spin_ = 0;
    // gen_ holds (generation << 16) | participants: a worker decides whether it takes part in a loop from the SAME atomic value it remembers as
    // `seen` (a separate plain `participants_` could be read after run() had already set up the next loop: a worker would then drain one generation
    // twice and run() could return while a body on its stack was still being executed -- ADVICE round 4)
    static constexpr unsigned long long PART_MASK = 0xffffull;
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void drain()
    {
        for (;;) {
            const long long b = next_.fetch_add(chunk_, std::memory_order_relaxed);
            if (b >= end_) break;
            (*body_)(b, b + chunk_ < end_ ? b + chunk_ : end_);
        }
    }
    void work(int idx)
    {
        inside() = true;
        unsigned long long seen = 0;
        for (;;) {
            bool got = false;
            for (int i = 0; i < spin_; ++i) {
                if (gen_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_relaxed)) {
                    got = true;
                    break;
                }
                relax();
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return stop_.load() || gen_.load() != seen; });
                sleepers_.fetch_sub(1);
            }
            if (stop_.load()) return;
            seen = gen_.load(std::memory_order_acquire);
            if (idx < (int)(seen & PART_MASK)) { // a participant cannot miss a generation: run() does not return before all of them have checked out
                drain();
                pending_.fetch_sub(1, std::memory_order_acq_rel);
            }
        }
    }

public:
    explicit Pool(int n)
    {
        const int hw = (int)std::thread::hardware_concurrency();
        spin_ = (hw > 0 && n <= hw) ? 20000 : 0; // oversubscribed: straight to sleep
        if (n > (int)PART_MASK) n = (int)PART_MASK;
        for (int t = 1; t < n; ++t) workers_.emplace_back([this, t] { work(t); });
    }
    ~Pool()
    {
        stop_.store(true);
        {
            std::lock_guard<std::mutex> lk(m_);
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(long long n, const std::function<void(long long, long long)>& body)
    {
        const long long blocks = (n + GRAIN - 1) / GRAIN;
        const int p = (int)(blocks < (long long)size() ? blocks : (long long)size()); // (size() <= PART_MASK: the constructor caps it)
        if (p <= 1) {
            inside() = true;
            body(0, n);
            inside() = false;
            return;
        }
        body_ = &body;
        end_ = n;
        chunk_ = (n + 8LL * p - 1) / (8LL * p);
        if (chunk_ < 1) chunk_ = 1;
        next_.store(0, std::memory_order_relaxed);
        pending_.store(p - 1, std::memory_order_relaxed);
        gen_.store((((gen_.load(std::memory_order_relaxed) >> 16) + 1) << 16) | (unsigned long long)p, std::memory_order_seq_cst); // (only run() writes gen_)
        if (sleepers_.load() > 0) {
            {
                std::lock_guard<std::mutex> lk(m_);
            }
            cv_.notify_all();
        }
        inside() = true;
        drain();
        inside() = false;
        while (pending_.load(std::memory_order_acquire) != 0) relax();
    }
};
inline Pool& pool()
{
    static Pool p(env_threads());
    return p;
}
inline bool parallel() { return env_threads() > 1 && limit() > 1 && !inside(); }
} // namespace detail_shim

template <class I, class F>
inline void parallel_for(I first, I last, I step, const F& f)
{
    if (!detail_shim::parallel() || !(first < last)) {
        for (I i = first; i < last; i += step) f(i);
        return;
    }
    const long long n = ((long long)last - (long long)first + (long long)step - 1) / (long long)step;
    const std::function<void(long long, long long)> body = [&](long long b, long long e) {
        for (long long k = b; k < e; ++k) f((I)((long long)first + k * (long long)step));
    };
    detail_shim::pool().run(n, body);
}
template <class I, class F>
inline void parallel_for(I first, I last, const F& f)
{
    parallel_for(first, last, (I)1, f);
}
template <class T>
class blocked_range {
    T b_, e_;

public:
    blocked_range(T b, T e, size_t = 1) : b_(b), e_(e) {}
    T begin() const { return b_; }
    T end() const { return e_; }
};
template <class T, class F>
inline void parallel_for(const blocked_range<T>& r, const F& f)
{
    if (!detail_shim::parallel() || !(r.begin() < r.end())) {
        f(r);
        return;
    }
    const long long n = (long long)r.end() - (long long)r.begin();
    const std::function<void(long long, long long)> body = [&](long long b, long long e) { f(blocked_range<T>((T)((long long)r.begin() + b), (T)((long long)r.begin() + e))); };
    detail_shim::pool().run(n, body);
}
} // namespace tbb
