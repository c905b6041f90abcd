// ORACLE / TEST INFRASTRUCTURE: see parallel_for.h
#pragma once
#include "parallel_for.h"
#include <cstddef>
namespace tbb {
class global_control {
public:
    enum parameter { max_allowed_parallelism, thread_stack_size };
    global_control(parameter p, size_t n)
    {
        if (p == max_allowed_parallelism) detail_shim::limit() = (int)n;
    }
};
} // namespace tbb
