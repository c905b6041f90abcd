// ORACLE / TEST INFRASTRUCTURE: see parallel_for.h
#pragma once
namespace tbb { namespace info { inline int default_concurrency() { return 1; } } }
