cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for v in s3 s1 j3 j1; do
echo "# $v: padded + shared at 3 workgroups per CU, then speed at the shipped residency"
ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_p$v.so python tools/nerf_fwd_residency.py 768 300 2>&1 | tail -1
ENERF_LIB_PATH=$R/enerf_amd/lib/variants/lib_$v.so python tools/bench_nerf_mlp.py 2>&1 | tail -1
done
