cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6r
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "whole_position or tuned_table or bench_batch or golden" > gpurun_out/r6r/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r6r/t1.log
tail -4 gpurun_out/r6r/t1.log
bash tools/final_profiles.sh r06
