set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | tee -a $O/rates.txt; }
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64"; do
  run A=0
  run RT_AMD_LIB=tools/diag/librt_amd_c64.so
  run RT_AMD_LIB=tools/diag/librt_amd_c1024.so
  run RT_WF_SLOTS=8000000
  run RT_WF_SLOTS=33000000
  run RT_WF_SLOTS=260000000
done
SCENE="final 100 3840 2160 256"; run A=0; run RT_WF_KERNEL=1
echo "== stamps" | tee -a $O/rates.txt
timeout -k 10 200 python3 tools/stamps.py tri_mesh 1920 1080 8 1000000 2>/dev/null | tee -a $O/stamps.txt
timeout -k 10 200 python3 tools/stamps.py final 1920 1080 16 100 2>/dev/null | tee -a $O/stamps.txt
