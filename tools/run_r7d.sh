cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7d
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r7d/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r7d/tests.log
tail -5 gpurun_out/r7d/tests.log
bash tools/final_profiles.sh r06
