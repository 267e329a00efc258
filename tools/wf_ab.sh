# A/B runs of the wavefront form: libraries under tools/diag (RT_AMD_LIB) and environment switches as arguments of run
run() { echo "== $*"; env "$@" timeout -k 10 150 python tools/wf_rate.py $SCENE 2>/dev/null | grep -o '"Grays_per_s": [0-9.]*'; }
for lib in tools/diag/librt_a.so tools/diag/librt_b.so; do
SCENE="tri_mesh 1000000 1920 1080 256"; run RT_AMD_LIB=$PWD/$lib
SCENE="final 100 1920 1080 64"; run RT_AMD_LIB=$PWD/$lib
done
