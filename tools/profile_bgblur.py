"""`-p bgblur:25` without `-b` (app/deepseg.cc:652-661) at 256 VGA streams, both forms, for rocprofv3:
   two-call = bsx_gaussian_blur_bgr into a per-stream background + bsx_step_batch; one-pass = bsx_step_batch_ex(BSX_STEP_BGBLUR(25)).
   usage: python tools/profile_bgblur.py two|one [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import backscrub_amd as bs  # noqa: E402
from tests.conftest import model_path  # noqa: E402

form, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, W, H = 256, 640, 480
mg = bs.MaskGen(model_path("lite"), W, H, n_streams=n)
fr = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda")
out, bl = torch.empty_like(fr), torch.empty_like(fr)
for _ in range(iters):
    if form == "two":
        mg.gaussian_blur(fr, 25, out=bl)
        mg.step(fr, bl, out)
    else:
        mg.step_ex(fr, None, out, bgblur=25)
torch.cuda.synchronize()
mg.close()
