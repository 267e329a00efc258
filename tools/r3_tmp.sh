set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for k in 4 1 2 3; do
echo "== RT_WF_KERNEL=$k" | tee -a $O/stamps.txt
RT_WF_KERNEL=$k timeout -k 10 200 python3 tools/stamps.py tri_mesh 1920 1080 8 1000000 2>/dev/null | tee -a $O/stamps.txt
RT_WF_KERNEL=$k timeout -k 10 200 python3 tools/stamps.py final 1920 1080 16 100 2>/dev/null | tee -a $O/stamps.txt
done
for s in "tri_mesh 1920 1080 8 1000000" "final 1920 1080 16 100"; do
RT_AMD_LIB=tools/diag/librt_amd_stampsrf.so RT_STAMPS_REFILL=1 RT_STAMPS_RF3=1 timeout -k 10 200 python3 tools/stamps.py $s 2>/dev/null | tee -a $O/stamps_refill.txt
done
