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
class Pool {
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cvStart_, cvDone_;
    const std::function<void(long long, long long)>* body_ = nullptr;
    std::atomic<long long> next_{ 0 };
    long long end_ = 0, chunk_ = 1;
    unsigned long long gen_ = 0;
    int active_ = 0;
    bool stop_ = false;
    void drain()
    {
        for (;;) {
            const long long b = next_.fetch_add(chunk_, std::memory_order_relaxed);
            if (b >= end_) break;
            (*body_)(b, b + chunk_ < end_ ? b + chunk_ : end_);
        }
    }

public:
    explicit Pool(int n)
    {
        for (int t = 1; t < n; ++t)
            workers_.emplace_back([this] {
                inside() = true;
                unsigned long long seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cvStart_.wait(lk, [&] { return stop_ || gen_ != seen; });
                        if (stop_) return;
                        seen = gen_;
                    }
                    drain();
                    {
                        std::lock_guard<std::mutex> lk(m_);
                        if (--active_ == 0) cvDone_.notify_one();
                    }
                }
            });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cvStart_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    void run(long long n, const std::function<void(long long, long long)>& body)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            body_ = &body;
            next_.store(0);
            end_ = n;
            const long long parts = 8LL * size();
            chunk_ = (n + parts - 1) / parts;
            if (chunk_ < 1) chunk_ = 1;
            active_ = (int)workers_.size();
            ++gen_;
        }
        cvStart_.notify_all();
        inside() = true;
        drain();
        inside() = false;
        std::unique_lock<std::mutex> lk(m_);
        cvDone_.wait(lk, [&] { return active_ == 0; });
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
