cd $GRAFT_REPO_ROOT
ENERF_LIB_PATH=$GRAFT_REPO_ROOT/enerf_amd/lib/variants/lib_tatime.so python tools/dev/ta_tiles.py 2>&1 | grep -v "^{" | tail -9
