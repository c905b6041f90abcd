mkdir -p gpurun_out
export IPCGPU_CCD_MODE=1
timeout 420 python -m pytest tests -m gpu -q -rf --tb=short --maxfail=15 -p no:cacheprovider > gpurun_out/r02c_gpu_tests_mode1.txt 2>&1
echo "mode1 rc=$?" >> gpurun_out/r02c_gpu_tests_mode1.txt
tail -5 gpurun_out/r02c_gpu_tests_mode1.txt
timeout 120 python __graft_entry__.py smoke > gpurun_out/r02c_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r02c_smoke.txt; tail -3 gpurun_out/r02c_smoke.txt
timeout 150 python bench.py > gpurun_out/r02c_bench_line.json 2> gpurun_out/r02c_bench_err.txt; echo "bench rc=$?"; head -c 600 gpurun_out/r02c_bench_line.json
if grep -q "failed" gpurun_out/r02c_gpu_tests_mode1.txt; then
  IPCGPU_CCD_MODE=0 timeout 300 python -m pytest tests -m gpu -q -rf --tb=line --maxfail=15 -p no:cacheprovider --deselect tests/test_gpu_vs_reference.py > gpurun_out/r02c_gpu_tests_mode0.txt 2>&1
  echo "mode0 rc=$?" >> gpurun_out/r02c_gpu_tests_mode0.txt; tail -5 gpurun_out/r02c_gpu_tests_mode0.txt
fi
