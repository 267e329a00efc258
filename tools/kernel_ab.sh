# A/B record of the four forms of the wavefront integrator (RT_WF_KERNEL): whole-call rates on the three BVH workloads,
# then one SQ counter pass per form on the 1M-triangle scene.  Usage: bash tools/kernel_ab.sh TAG  ->  gpurun_out/TAG/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_kernel_ab}; mkdir -p $O
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "cornell_cluster 0 1920 1080 64"; do
  for k in 4 1 2 3; do
    echo "RT_WF_KERNEL=$k $(RT_WF_KERNEL=$k timeout -k 10 150 python3 tools/wf_rate.py $SCENE 2>/dev/null)" | tee -a $O/rates.txt
  done
done
for k in 4 1 2 3; do
  RT_WF_KERNEL=$k timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq_k$k -- python3 tools/wf_rate.py tri_mesh 1000000 1920 1080 64 > $O/sq_k$k.log 2>&1
  RT_WF_KERNEL=$k timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm_k$k -- python3 tools/wf_rate.py tri_mesh 1000000 1920 1080 64 > $O/grbm_k$k.log 2>&1
done
echo done
