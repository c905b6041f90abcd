// Host build of ipc_amd/csrc/orient3d_exact.h for tests/test_orient3d.py (test infrastructure: the product compiles the same header with hipcc).
#include "../../ipc_amd/csrc/orient3d_exact.h"
extern "C" int o3_orient3d(const double* p12) { return ipcgpu::o3::orient3d(p12, p12 + 3, p12 + 6, p12 + 9); }
extern "C" int o3_orient3d_exact(const double* p12) { return ipcgpu::o3::orient3d_exact(p12, p12 + 3, p12 + 6, p12 + 9); }
