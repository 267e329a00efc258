# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench command, then PMC passes
# (each in a run of its own) for the Cornell megakernel and for the wavefront kernels on the triangle scene.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r02d}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 bench.py --no-cpu-baseline --no-extras > $O/trace_default.log 2>&1
C="python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cornell_fetch -- $C > $O/cornell_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cornell_write -- $C > $O/cornell_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/cornell_sq -- $C > $O/cornell_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $O/cornell_sq2 -- $C > $O/cornell_sq2.log 2>&1
T="python3 bench.py --no-cpu-baseline --workload tri_mesh --steps 1 --warmup 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tri -- $T > $O/trace_tri.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/tri_fetch -- $T > $O/tri_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/tri_write -- $T > $O/tri_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/tri_tcc -- $T > $O/tri_tcc.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/tri_sq -- $T > $O/tri_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/tri_grbm -- $T > $O/tri_grbm.log 2>&1
echo done
