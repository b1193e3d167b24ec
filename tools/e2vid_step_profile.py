"""Per-kernel time of one E2VIDRecurrent time step at the TRAINING shape (12 x 5 x 128 x 128 voxels per step): run under
tools/profile_py.sh (rocprofv3 kernel stats).  usage: bash tools/profile_py.sh tools/e2vid_step_profile.py [B H W steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd.unet import E2VIDRecurrent  # noqa: E402

b, h, w, steps = (int(v) for v in (sys.argv[1:5] + ["12", "128", "128", "40"][len(sys.argv) - 1:]))
torch.manual_seed(0)
net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                          num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
ev = torch.round(torch.randn((b, steps, 5, h, w), device="cuda") * 2)
sc = torch.ones((b, 2), device="cuda") * 3
with torch.no_grad():
    for rep in range(3):
        net.reset_states()
        for t in range(steps):
            img = net(ev[:, t], sc)["image"]
torch.cuda.synchronize()
print("done", tuple(img.shape))
