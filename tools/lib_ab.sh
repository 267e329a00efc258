# A/B of variant libraries (make variant NAME=x EXTRA=...): whole-call rates of the BVH workloads per library.
# Usage: bash tools/lib_ab.sh TAG name1 name2 ...   (name "-" = the product library)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "cornell_cluster 0 1920 1080 64"; do
  for n in "$@"; do
    if [ "$n" = "-" ]; then L=cuda-raytracer_amd/librt_amd.so; else L=tools/diag/librt_amd_$n.so; fi
    echo "$n $(RT_AMD_LIB=$L timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null)" | tee -a $O/rates.txt
  done
done
