set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python3 bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-300
