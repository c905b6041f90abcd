mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -q -rf --tb=short --maxfail=15 -p no:cacheprovider > gpurun_out/r02f_gpu_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r02f_gpu_tests.txt
tail -8 gpurun_out/r02f_gpu_tests.txt
timeout 60 python tools/_gpu_diag.py > gpurun_out/r02f_scene_counts.txt 2>&1
tail -8 gpurun_out/r02f_scene_counts.txt | cut -c1-250
