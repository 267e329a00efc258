set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>&1 | tail -3 | tee -a $O/rates.txt; }
SCENE="tri_mesh 20000 640 360 16"; run RT_WF_KERNEL=1; run RT_WF_KERNEL=4 RT_WF_DEBUG=1
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "cornell_cluster 0 1920 1080 64"; do
  run RT_WF_KERNEL=1; run RT_WF_KERNEL=4
done
RT_WF_KERNEL=4 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_k4.log 2>&1 || { tail -30 $O/pytest_k4.log; exit 1; }
tail -2 $O/pytest_k4.log
