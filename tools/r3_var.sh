# rates of variant libraries (tools/diag/librt_amd_NAME.so) beside the in-tree one: bash tools/r3_var.sh TAG NAME...
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | tee -a $O/rates.txt; }
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64"; do
  run RT_WF_KERNEL=1
  run RT_WF_KERNEL=2
  for v in "$@"; do run RT_WF_KERNEL=2 RT_AMD_LIB=$PWD/tools/diag/librt_amd_$v.so; done
done
