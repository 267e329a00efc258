# A/B runs of the wavefront form: environment switches (RT_AMD_LIB=... selects a library) as arguments of run
run() { echo "== $*"; env "$@" timeout -k 10 150 python tools/wf_rate.py $SCENE 2>/dev/null | grep -o '"Grays_per_s": [0-9.]*'; }
SCENE="tri_mesh 1000000 1920 1080 256"; run A=0
SCENE="final 100 1920 1080 64"; run A=0
SCENE="final 100 3840 2160 256"; run A=0
