# the round-end sequence on the GPU box: every -m gpu test in one process, then the default bench.py run.
# Usage (from the container): gpurun --timeout 1200 -- 'bash tools/gpu_full.sh TAG'   ->  gpurun_out/TAG/
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-full}; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 900 python3 bench.py 2>$O/bench.err > $O/bench.json
cut -c1-200 $O/bench.json; grep done $O/bench.err
