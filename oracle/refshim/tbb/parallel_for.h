// ORACLE / TEST INFRASTRUCTURE: serial stand-in for oneTBB's parallel_for (same results, one thread), so that the
// reference's own sources compile here without oneTBB.  See mini_eigen.hpp.
#pragma once
#include <cstddef>
namespace tbb {
template <class I, class F>
inline void parallel_for(I first, I last, I step, const F& f)
{
    for (I i = first; i < last; i += step) f(i);
}
template <class I, class F>
inline void parallel_for(I first, I last, const F& f)
{
    for (I i = first; i < last; ++i) f(i);
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
inline void parallel_for(const blocked_range<T>& r, const F& f) { f(r); }
} // namespace tbb
