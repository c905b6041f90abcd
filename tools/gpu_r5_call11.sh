#!/bin/bash
# Round 5, eleventh GPU call: where the time of a single-workgroup front goes (ipc_amd/libipcgpu_probe.so = the library with -DMF_FUSED_PROBE: thread 0 of every fused
# front stamps its phases with the 100 MHz wall clock; level averages printed once).  mat150 and the contact stack's matrix.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r5c11
mkdir -p $out
IPCGPU_LIB_VARIANT=probe timeout 200 python bench.py --no-cpu-baseline --no-contact --no-large --steps 20 --warmup 5 2>&1 >/dev/null | grep "fused probe" | tee $out/probe_mat150.txt
IPCGPU_LIB_VARIANT=probe timeout 200 python bench.py --no-cpu-baseline --no-contact --no-large --size 433 --steps 4 --warmup 2 2>&1 >/dev/null | grep "fused probe" | tee $out/probe_mat433.txt
