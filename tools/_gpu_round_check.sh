mkdir -p gpurun_out
timeout 70 python -m pytest tests/test_gpu_vs_reference.py -m gpu -q -rf --tb=short -p no:cacheprovider -k "point_triangle or cubes_dhat" > gpurun_out/r02g_gpu_new_scene_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r02g_gpu_new_scene_tests.txt
tail -6 gpurun_out/r02g_gpu_new_scene_tests.txt | cut -c1-300
