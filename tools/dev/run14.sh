cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh k_grid_fwd $GRAFT_REPO_ROOT/tools/dev/upd_full.py 2>&1 | tail -40
