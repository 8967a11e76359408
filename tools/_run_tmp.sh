cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7j
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r7j/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r7j/tests.log
tail -4 gpurun_out/r7j/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
