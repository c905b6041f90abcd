// ORACLE / TEST INFRASTRUCTURE: libigl is un-vendored; the file I/O and 2-D parametrisation helpers the reference
// includes here are not on the Newton path.  Calls compile and do nothing.
#pragma once
#include <string>
namespace igl {
template <class... A> inline bool list_to_matrix(const A&...) { return false; }
} // namespace igl
