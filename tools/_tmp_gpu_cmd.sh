python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_final.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_final.txt
bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1; echo "final rc=$?"
