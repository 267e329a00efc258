set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | tee -a $O/rates.txt; }
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64"; do
  run RT_WF_KERNEL=1; run RT_WF_KERNEL=3
done
KERNELS=3 bash tools/r3_stamps.sh $1
RT_WF_KERNEL=3 timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_k3.log 2>&1 || { tail -30 $O/pytest_k3.log; exit 1; }
tail -2 $O/pytest_k3.log
