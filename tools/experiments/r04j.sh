R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_real_frame.py -q -x > $O/r04j_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04j_tests.log)"
TESTS="tests/test_gpu_pipeline.py" bash tools/experiments/run_variants.sh 2>&1 | grep -v "tests rc" 
timeout -k 10 600 python tools/stress_parity.py 1200 41 > $O/r04j_stress_parity.log 2>&1; echo "stress_parity rc=$? $(tail -2 $O/r04j_stress_parity.log)"
timeout -k 10 400 python tools/stress_jpeg.py 800 7 > $O/r04j_stress_jpeg.log 2>&1; echo "stress_jpeg rc=$? $(tail -2 $O/r04j_stress_jpeg.log)"
