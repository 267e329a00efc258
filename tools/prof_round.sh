# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench command, then PMC passes
# (each in a run of its own) for every line the default bench.py run prints: the Cornell megakernel (headline and
# the 64x64 smoke frame), the wavefront kernels on the triangle scene and on the final scene (the latter at
# 1920x1080 x 64 spp: the full configs[4] frame is a minute a pass and the per-ray shares do not depend on the frame).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03}
mkdir -p $O
# PART=1: default command, Cornell, triangle scene; PART=2: smoke, final scene, rows per ray (a gpurun call is 20 minutes at most)
if [ "${PART:-1}" = 1 ]; then
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
echo part 1 done; exit 0
fi
C="python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0"
S="python3 bench.py --no-cpu-baseline --workload cornell_smoke --steps 1 --warmup 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/smoke_fetch -- $S > $O/smoke_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/smoke_write -- $S > $O/smoke_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/smoke_sq -- $S > $O/smoke_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/smoke_grbm -- $S > $O/smoke_grbm.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/cornell_grbm -- $C > $O/cornell_grbm.log 2>&1
F="python3 bench.py --no-cpu-baseline --workload final --width 1920 --height 1080 --spp 64 --steps 1 --warmup 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_final -- $F > $O/trace_final.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/final_fetch -- $F > $O/final_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/final_write -- $F > $O/final_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/final_tcc -- $F > $O/final_tcc.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/final_sq -- $F > $O/final_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/final_grbm -- $F > $O/final_grbm.log 2>&1
# rows per ray of the two wavefront scenes from the diagnostic build (same frame sizes, fewer samples)
timeout -k 10 200 python3 tools/stamps.py tri_mesh 1920 1080 8 1000000 > $O/stamps_tri.txt 2>/dev/null
timeout -k 10 200 python3 tools/stamps.py final 1920 1080 16 100 > $O/stamps_final.txt 2>/dev/null
echo done
