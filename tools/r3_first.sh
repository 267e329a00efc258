# round 3, first GPU call: GPU suite, strong-scaling projection at HEAD, cooperative-gather ubench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3a/pytest.log 2>&1 && tail -3 gpurun_out/r3a/pytest.log
timeout -k 10 200 python3 tools/strong_projection.py r03a --tiles > gpurun_out/r3a/proj.log 2>&1 && tail -8 gpurun_out/r3a/proj.log
for mb in 2 110; do timeout -k 10 120 tools/ubench/gather --mb $mb --iters 1000 >> gpurun_out/r3a/gather.jsonl; done
cut -c1-220 gpurun_out/r3a/gather.jsonl
