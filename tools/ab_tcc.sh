# L2 counters of the wavefront kernels on the triangle scene with the environment given as arguments: $1 = tag
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tcc_$1
mkdir -p $O
T="python3 tools/wf_rate.py tri_mesh 1000000 1920 1080 64"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum --output-format csv -d $O/tcc -- $T > $O/tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $T > $O/fetch.log 2>&1
for d in tcc fetch; do python3 tools/pmc_summary.py $O/$d wf_intersect wf_logic; done
