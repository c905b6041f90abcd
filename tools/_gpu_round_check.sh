# what the round's GPU evidence under profiles/ was produced with (one gpurun call each):
#   python -m pytest tests -m gpu -q -rf --tb=short          -> profiles/r02f_gpu_tests.txt (+ r02g_gpu_new_scene_tests.txt)
#   python tools/_gpu_diag.py                                 -> profiles/r02f_scene_counts_gpu_vs_reference.txt
#   rocprofv3 --kernel-trace --stats -- python bench.py ...   -> profiles/r02f_kernel_stats.md (tools/rocprof_summary.py)
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -q -rf --tb=short --maxfail=15 -p no:cacheprovider > gpurun_out/gpu_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/gpu_tests.txt
tail -8 gpurun_out/gpu_tests.txt
timeout 120 python tools/_gpu_diag.py > gpurun_out/scene_counts.txt 2>&1
