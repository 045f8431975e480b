cd $GRAFT_REPO_ROOT
bash tools/update_timeline.sh 2>&1 | tail -40
