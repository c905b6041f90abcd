// ORACLE / TEST INFRASTRUCTURE: see parallel_for.h
#pragma once
#include "parallel_for.h"
namespace tbb { namespace info { inline int default_concurrency() { return detail_shim::env_threads(); } } }
