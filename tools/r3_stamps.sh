set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for k in ${KERNELS:-1 2}; do
echo "== kernel $k" | tee -a $O/stamps.txt
RT_WF_KERNEL=$k timeout -k 10 200 python3 tools/stamps.py tri_mesh 1920 1080 8 1000000 2>/dev/null | tee -a $O/stamps.txt
RT_WF_KERNEL=$k timeout -k 10 200 python3 tools/stamps.py final 1920 1080 16 100 2>/dev/null | tee -a $O/stamps.txt
done
