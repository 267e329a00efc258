import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools/diag/librt_amd_dbg.so")
spec = importlib.util.spec_from_file_location("rtb", os.path.join(ROOT, "cuda-raytracer_amd/binding.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
import torch
W, H = 96, 64
sc = B.Scene(sys.argv[1] if len(sys.argv) > 1 else "cornell", W, H); ctx = B.Context(0); ctx.upload(sc)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device='cuda'); torch.cuda.synchronize()
ctx.render_accumulate(acc, W, H, 0, 1, 419, 3, sc.background); ctx.synchronize()
print(acc.view(H, W, 3)[2, 22])
