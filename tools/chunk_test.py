import sys, json
sys.path.insert(0, '/root/repo')
import os
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import __graft_entry__ as e, torch
b=e.load_binding(); ctx=b.Context(0)
W,H=3840,2160
sc=b.Scene("final",W,H,100); ctx.upload(sc)
acc=torch.zeros(H*W*3,dtype=torch.float32,device="cuda"); torch.cuda.synchronize()
ctx.render_accumulate(acc,W,H,0,16,419,100,sc.background); ctx.synchronize()
for total,chunk in ((512,512),(512,128),(512,64),(512,256)):
    ctx.reset_counters()
    for s0 in range(0,total,chunk):
        ctx.render_accumulate(acc,W,H,1000+s0,1000+s0+chunk,419,100,sc.background)
    c=ctx.counters()
    print(json.dumps({"total":total,"chunk":chunk,"Grays_per_s":c.rays/(c.render_ms*1e-3)/1e9,"ms":c.render_ms}),flush=True)
