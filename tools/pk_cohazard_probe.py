"""Packed-float32 instructions under co-scheduling with matrix-core kernels (round 5 finding): a victim kernel launched on the current stream
while a disturber runs on a second stream, compared with the victim's result when it runs alone.

    python tools/pk_cohazard_probe.py [path to an alternative libv2v_hip.so]     (e.g. a build with -DV2V_UPSAMPLE_ATTR= : packed blends)

Victims: the x2 upsampling kernel; the ESIM headline instance (float32 clips, bilinear bins: v_pk_fma_f32 accumulation) and a uint8 SUM
instance; torch's own elementwise kernel.  Disturbers: this library's ConvLSTM step (v_mfma_f32_32x32x16_bf16 + global_load_lds_dwordx4),
its halo convolution, rocBLAS bf16 GEMM (torch.mm), a float32 elementwise stream."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["V2V_HIP_LIB"] = sys.argv[1]
from v2v_amd import _lib, convlstm as CL, esim  # noqa: E402

print("library:", getattr(_lib, "LIB_PATH", None) or os.environ.get("V2V_HIP_LIB") or "v2v_amd/libv2v_hip.so")
g = torch.Generator().manual_seed(1)
side = torch.cuda.Stream()


def lstm(c, hw):
    xx = torch.randn((12, hw, hw, c), generator=g).bfloat16().cuda()
    hp = torch.randn((12, hw, hw, c), generator=g).bfloat16().cuda()
    cp = torch.randn((12, hw, hw, c), generator=g).cuda()
    packed = CL.pack_gate_weights((torch.randn((4 * c, 2 * c, 3, 3), generator=g) * 0.02).cuda())
    bias = torch.zeros(4 * c).cuda()
    return lambda: CL.convlstm_step(xx, hp, cp, packed, bias, nchw_dtype=None)


def halo():
    x = torch.randn((12, 128, 128, 64), generator=g).bfloat16().cuda()
    packed = CL.pack_conv_weights((torch.randn((32, 64, 5, 5), generator=g) * 0.03).cuda())
    bias = torch.zeros(32).cuda()
    return lambda: CL.conv_nhwc(x, packed, bias, 5, relu=True)


def mm():
    a = torch.randn((2048, 2048), device="cuda").bfloat16()
    return lambda: torch.mm(a, a)


def elementwise():
    a = torch.randn((64, 1024, 1024), device="cuda")
    return lambda: a * 1.5 + 2.0


ux, usk = torch.randn((12, 64, 64, 64), generator=g).bfloat16().cuda(), torch.randn((12, 64, 64, 64), generator=g).bfloat16().cuda()
clips_f32 = esim.synth_clips(64, 32, 256, 256, dtype=torch.float32)
clips_u8 = esim.synth_clips(48, 41, 256, 256, dtype=torch.uint8)
P = [0.2, 0.2, 0.1, 0.001, 0.1]
from v2v_amd import v2e  # noqa: E402
vparams = v2e.make_params(24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1)
victims = {
    "upsample2x (64 ch @64^2)": lambda: CL.upsample2x_nhwc(ux, usk),
    "esim f32 bilinear (v_pk_fma_f32)": lambda: esim.esim_voxel_batch(clips_f32, P, bin_mode="bilinear", num_bins=5, seed=3),
    "esim u8 sum": lambda: esim.esim_voxel_batch(clips_u8, [0.2, 0.3, 0.05, 5e-4, 1.0], bin_mode="sum", num_bins=5, seed=3),
    "esim f32 bilinear, 2-pixel mapping": lambda: esim.esim_voxel_batch(clips_f32[:24], [0.2, 0.3, 0.1, 0.001, 0.1], bin_mode="bilinear", num_bins=5, seed=3, mapping="2px"),
    "v2e f32 (shot noise, leak)": lambda: v2e.v2e_voxel_batch(clips_f32[:32], vparams, bin_mode="bilinear", num_bins=5, rng_mode="philox", seed=3),
    "v2e u8 (shot noise, leak)": lambda: v2e.v2e_voxel_batch(clips_u8[:32], vparams, bin_mode="sum", num_bins=5, rng_mode="philox", seed=3),
    "torch elementwise": lambda: ux.float() * 0.75 + usk.float() * 0.25,
}
disturbers = {"convlstm_step 64@64^2": lstm(64, 64), "convlstm_step 256@16^2 (KS=2)": lstm(256, 16), "conv_halo 64->32 @128^2": halo(),
              "rocBLAS bf16 mm 2048^3": mm(), "torch elementwise": elementwise()}
torch.cuda.synchronize()
for vname, victim in victims.items():
    solo = victim()
    torch.cuda.synchronize()
    again = sum(not torch.equal(victim(), solo) for _ in range(10))
    row = [f"alone {again}/10"]
    for dname, dist in disturbers.items():
        bad = worst = 0
        for rep in range(int(os.environ.get("PK_PROBE_ROUNDS", "8"))):
            with torch.cuda.stream(side):
                for _ in range(24):
                    dist()
            outs = [victim() for _ in range(4)]
            torch.cuda.synchronize()
            for o in outs:
                n = int((o != solo).sum())
                bad += n > 0
                worst = max(worst, n)
        row.append(f"{dname}: {bad}/{4 * int(os.environ.get('PK_PROBE_ROUNDS', '8'))} launches differ (worst {worst} elements)")
    print(f"{vname:34s} | " + " | ".join(row))
