cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for v in pwe pad100 pwe; do
echo "# $v"
ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_$v.so python tools/nerf_fwd_residency.py 768 300 2>&1 | tail -1
done
