# A/B runs of the wavefront form (environment switches given as arguments of run)
run() { echo "== $*"; env "$@" timeout -k 10 150 python tools/wf_rate.py $SCENE 2>/dev/null | grep Grays; }
SCENE="tri_mesh 1000000 1920 1080 64"; run A=0
SCENE="tri_mesh 1000000 1920 1080 256"; run A=0; run RT_WF_PART=0; run RT_WF_PART=0 RT_WF_AFFINE=1
SCENE="final 100 1920 1080 64"; run A=0
SCENE="final 100 3840 2160 512"; run A=0; run RT_WF_PART=0; run RT_WF_PART=0 RT_WF_AFFINE=1
