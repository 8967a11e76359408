cd $GRAFT_REPO_ROOT
export B=64
bash tools/trace_timeline.sh
cat gpurun_out/r3t/timeline_graph_b64.txt
