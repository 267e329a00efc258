set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01b
mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline > $O/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_sq2.log 2>&1
cat $O/bench.json
