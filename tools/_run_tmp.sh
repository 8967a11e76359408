cd $GRAFT_REPO_ROOT
bash tools/final_profiles.sh r06
