set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01tri
mkdir -p $O
CMD="python3 bench.py --no-cpu-baseline --workload tri_mesh --spp 16 --steps 1 --warmup 0"
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tcc -- $CMD > $O/pmc_tcc.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- $CMD > $O/pmc_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tcp -- $CMD > $O/pmc_tcp.log 2>&1 || true
echo done
