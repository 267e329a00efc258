# A/B runs of the wavefront form (environment switches given as arguments of run)
run() { echo "== $*"; env "$@" timeout -k 10 150 python tools/wf_rate.py $SCENE 2>/dev/null | grep -o '"Grays_per_s": [0-9.]*'; }
SCENE="tri_mesh 1000000 1920 1080 256"; for r in 16 24 32 40 48; do run RT_WF_REFILL=$r; done
SCENE="final 100 1920 1080 64"; for r in 16 24 32 40 48; do run RT_WF_REFILL=$r; done
