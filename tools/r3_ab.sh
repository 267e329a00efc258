# A/B of the two intersect kernels: parity tests that run the wavefront form, then rates (RT_WF_KERNEL=1|2)
# usage: bash tools/r3_ab.sh TAG [quick]
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== $*" | tee -a $O/rates.txt; env "$@" timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null | tee -a $O/rates.txt; }
SCENE="tri_mesh 1000000 1920 1080 256"; run RT_WF_KERNEL=1; run RT_WF_KERNEL=2
SCENE="final 100 1920 1080 64"; run RT_WF_KERNEL=1; run RT_WF_KERNEL=2
if [ "$2" != "quick" ]; then
SCENE="tri_mesh 1000000 1920 1080 256"; for r in 4 8 16 24; do run RT_WF_KERNEL=2 RT_WF_REFILL=$r; done
SCENE="final 100 1920 1080 64"; for r in 4 8 16 24; do run RT_WF_KERNEL=2 RT_WF_REFILL=$r; done
SCENE="cornell_cluster 0 1920 1080 64"; run RT_WF_KERNEL=1; run RT_WF_KERNEL=2
fi
