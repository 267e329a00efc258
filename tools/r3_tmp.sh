set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "forms_give or special_rays or failure_leaves" 2>&1 | tail -15
