// Host build of ipc_amd/csrc/jacobi9_device.h for tests/test_jacobi9.py (test infrastructure: the product compiles the same header with hipcc).
#include "../../ipc_amd/csrc/jacobi9_device.h"
extern "C" int j9_make_pd(int nn, double* A) { return ipcgpu::j9::make_pd_stencil_reg(nn, A); }
