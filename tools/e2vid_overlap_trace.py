"""One hipGraph replay of E2VIDRecurrent.forward_sequence at the training shape under rocprofv3 --kernel-trace: run as
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -o t -- python3 $REPO/tools/e2vid_overlap_trace.py
then `python tools/e2vid_overlap_trace.py --analyze /tmp/ovl` prints, for the LAST replay: wall time, summed kernel time, the time at least
one / two / three kernels were resident, and the per-kernel-family totals."""
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    rows = [r for r in rows if "v2v::" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = int(os.environ.get("PER_REPLAY", "0")) or len(rows) // 6
    last = rows[-n:]
    t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in last] + [(int(r["End_Timestamp"]), -1) for r in last])
    depth, prev, occ = 0, t0, {}
    for t, d in ev:
        occ[depth] = occ.get(depth, 0) + (t - prev)
        depth, prev = depth + d, t
    wall = (t1 - t0) / 1e3
    print(f"kernels in the replay: {len(last)}   wall {wall:.1f} us   summed kernel time {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e3:.1f} us")
    for k in sorted(occ):
        print(f"  {k} kernels resident: {occ[k] / 1e3:9.1f} us  ({occ[k] / 1e3 / wall:5.1%})")
    fam = {}
    for r in last:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        fam.setdefault(name, [0, 0.0])
        fam[name][0] += 1
        fam[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for name, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {us:9.1f} us  {c:5d} x {us / c:7.1f}  {name}")
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd.unet import E2VIDRecurrent  # noqa: E402

torch.manual_seed(0)
net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                          num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
B, T, HW = (int(v) for v in os.environ.get("OVL_SHAPE", "12,40,128").split(","))    # config 5: OVL_SHAPE=8,8,256
ev = torch.round(torch.randn((B, T, 5, HW, HW), device="cuda") * 2)
sc = torch.ones((B, 2), device="cuda") * 3
with torch.no_grad():
    for _ in range(4):
        net.forward_sequence(ev, sc, graph=True)
torch.cuda.synchronize()
print("done")
