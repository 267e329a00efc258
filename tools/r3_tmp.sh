set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | tee -a $O/rates.txt; }
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "cornell_cluster 0 1920 1080 64" "random_spheres 0 1920 1080 64"; do
  run A=0
  run RT_AMD_LIB=tools/diag/librt_amd_flat.so
  run A=1
done
