# A/B runs of the wavefront form: environment switches (RT_AMD_LIB=... selects a library) as arguments of run
run() { echo "== $*"; env "$@" timeout -k 10 150 python tools/wf_rate.py $SCENE 2>/dev/null | grep -o '"Grays_per_s": [0-9.]*'; }
for SCENE in "tri_mesh 1000000 1920 1080 256" "final 100 1920 1080 64" "final 100 3840 2160 256" "cornell_cluster 0 1024 1024 64"; do
echo "#### $SCENE"
run A=0
run RT_AMD_LIB=$PWD/tools/diag/librt_w5.so
run RT_AMD_LIB=$PWD/tools/diag/librt_w6.so
done
