# lib_ab5.sh over the three wavefront scenes only.  Usage: bash tools/lib_ab3.sh TAG name1 name2 ...
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; rm -f $O/rates.txt; shift
for rep in 1 2; do
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "cornell_cluster 0 1920 1080 64"; do
  for n in "$@"; do
    if [ "$n" = "-" ]; then L=cuda-raytracer_amd/librt_amd.so; else L=tools/diag/librt_amd_$n.so; fi
    echo "$n $(RT_AMD_LIB=$L timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | cut -c1-120)" | tee -a $O/rates.txt
  done
done; done
