cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
echo "# shipped library, 3000 launches per cap"
for CAP in 512 768; do python tools/nerf_fwd_residency.py $CAP 3000 | tail -1; done
V3=$(bash tools/dev/build_variant.sh barpad nerf_mlp.hip "-mllvm -amdgpu-mfma-padding-ratio=100" | tail -1)
echo "# barrier in, MFMAs padded apart, 3000 launches per cap"
for CAP in 512 768; do ENERF_LIB_PATH=$V3 python tools/nerf_fwd_residency.py $CAP 3000 | tail -1; done
rm -f $V3
echo "# both nets per launch, 2000 repeats through both directions"
python tools/bench_nerf_mlp.py --runs 2000 | tail -1
python tools/soak.py 20000 2>&1 | tail -4
