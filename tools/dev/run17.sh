cd $GRAFT_REPO_ROOT
ENERF_EXPERIMENT_NO_MARCH_WAIT=1 bash tools/r05_timeline.sh r05p default | grep -v "^{"
