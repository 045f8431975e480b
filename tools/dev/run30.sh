cd $GRAFT_REPO_ROOT
bash tools/update_timeline.sh --strong-rays 0 --other-legs 0 2>&1 | tail -45
