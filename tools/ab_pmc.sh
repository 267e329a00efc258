# Counter passes of tools/wf_rate.py: $1 = output tag, the rest = wf_rate.py's arguments (default: the triangle scene); RT_AMD_LIB selects the library
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_$1
shift
mkdir -p $O
T="python3 tools/wf_rate.py ${*:-tri_mesh 1000000 1920 1080 64}"
timeout -k 10 100 $T > $O/rate.json 2>/dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -- $T > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- $T > $O/sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum --output-format csv -d $O/tcc -- $T > $O/tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $T > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $T > $O/write.log 2>&1
cat $O/rate.json
for d in sq sq2 tcc fetch write; do python3 tools/pmc_summary.py $O/$d wf_intersect wf_logic; done
