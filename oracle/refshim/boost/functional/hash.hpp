// ORACLE / TEST INFRASTRUCTURE: the one function of boost/functional/hash.hpp the reference's containers use.
#pragma once
#include <cstddef>
#include <functional>
namespace boost {
template <class T>
inline void hash_combine(std::size_t& seed, const T& v)
{
    seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
template <class It>
inline std::size_t hash_range(It b, It e)
{
    std::size_t s = 0;
    for (; b != e; ++b) hash_combine(s, *b);
    return s;
}
} // namespace boost
