"""Kernels of the library under CO-SCHEDULING with matrix-core work on another stream (round 5): a data simulator runs beside a training
loop's GEMMs, so every kernel must give its stand-alone result whatever shares the CU.  The round-4 library did not: packed float32
instructions (v_pk_mul/add/fma_f32) returned wrong values in lanes 48-63 while this library's ConvLSTM step or a rocBLAS bf16 GEMM ran on
a second stream -- the x2 upsampling kernel on nearly every launch, the v2e simulator on some (tools/pk_cohazard_probe.py,
profiles/r05/pk_cohazard_probe.txt).  The library is now built without those instructions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _disturbers():
    from v2v_amd import convlstm as CL
    g = torch.Generator().manual_seed(5)
    c, hw = 64, 64
    xx = torch.randn((12, hw, hw, c), generator=g).bfloat16().cuda()
    hp = torch.randn((12, hw, hw, c), generator=g).bfloat16().cuda()
    cp = torch.randn((12, hw, hw, c), generator=g).cuda()
    packed = CL.pack_gate_weights((torch.randn((4 * c, 2 * c, 3, 3), generator=g) * 0.02).cuda())
    bias = torch.zeros(4 * c).cuda()
    a = torch.randn((2048, 2048), device="cuda").bfloat16()
    return {"convlstm_step": lambda: CL.convlstm_step(xx, hp, cp, packed, bias, nchw_dtype=None), "rocblas_bf16_mm": lambda: torch.mm(a, a)}


def _victims():
    import numpy as np
    from v2v_amd import convlstm as CL, esim, frontend, loader, postops, v2e
    g = torch.Generator().manual_seed(6)
    ux, usk = torch.randn((12, 64, 64, 64), generator=g).bfloat16().cuda(), torch.randn((12, 64, 64, 64), generator=g).bfloat16().cuda()
    f32 = esim.synth_clips(32, 32, 256, 256, dtype=torch.float32)
    u8 = esim.synth_clips(32, 41, 256, 256, dtype=torch.uint8)
    train = esim.synth_clips(12, 201, 128, 128, dtype=torch.uint8)
    raw = torch.randint(0, 256, (6, 44, 360, 640, 3), generator=g, dtype=torch.uint8).cuda()
    table = np.array([[10 * i, 30 * i, 200 + 20 * i, i & 1] for i in range(6)], dtype=np.int32)
    idx = np.tile(np.arange(41, dtype=np.int32), (6, 1))
    grid = torch.round(torch.randn((12, 40, 5, 120, 120), generator=g) * 3).cuda()
    vp = v2e.make_params(24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1)
    return {
        "upsample2x": lambda: CL.upsample2x_nhwc(ux, usk),
        "esim_f32_bilinear": lambda: esim.esim_voxel_batch(f32, [0.2, 0.2, 0.1, 0.001, 0.1], bin_mode="bilinear", num_bins=5, seed=3),
        "esim_f32_2px": lambda: esim.esim_voxel_batch(f32[:24], [0.2, 0.3, 0.1, 0.001, 0.1], bin_mode="bilinear", num_bins=5, seed=3, mapping="2px"),
        "esim_u8_sum": lambda: esim.esim_voxel_batch(u8, [0.2, 0.3, 0.05, 5e-4, 1.0], bin_mode="sum", num_bins=5, seed=3),
        "v2e_u8": lambda: v2e.v2e_voxel_batch(u8, vp, bin_mode="sum", num_bins=5, rng_mode="philox", seed=3),
        "esim_u8_1px_quad_shared_noise": lambda: esim.esim_voxel_batch(train, [0.2, 0.3, 0.05, 5e-4, 1.0], bin_mode="sum", num_bins=5, seed=3, mapping="1px"),
        "frontend_720p_crop_resize_gray": lambda: frontend.prepare_clips_batch(raw, table, idx, 128, "gray")[1],
        "normalize_and_pad": lambda: postops.normalize_and_pad(grid, normalize=True),
        "clip_frames_f32": lambda: loader.clip_frames_f32(u8[:, :, :, :, None], list(range(1, 41, 5))),
        "v2e_f32": lambda: v2e.v2e_voxel_batch(f32, vp, bin_mode="bilinear", num_bins=5, rng_mode="philox", seed=3),
    }


@pytest.mark.parametrize("victim", ["upsample2x", "esim_f32_bilinear", "esim_f32_2px", "esim_u8_sum", "v2e_u8", "v2e_f32", "esim_u8_1px_quad_shared_noise",
                                    "frontend_720p_crop_resize_gray", "normalize_and_pad", "clip_frames_f32"])
def test_kernel_results_do_not_depend_on_what_shares_the_cu(victim):
    run = _victims()[victim]
    side = torch.cuda.Stream()
    solo = run()
    torch.cuda.synchronize()
    for name, disturb in _disturbers().items():
        for rep in range(6):
            with torch.cuda.stream(side):
                for _ in range(24):
                    disturb()
            outs = [run() for _ in range(4)]
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o, solo), f"{victim} differs from its stand-alone result while {name} runs on another stream ({int((o != solo).sum())} elements)"


def test_entry_points_from_several_host_threads():
    """The C ABI from several HOST threads at once (ctypes drops the GIL for the call), each on its own HIP stream: a data-loader thread next
    to a training thread is the normal deployment.  Status text is thread-local, launch attributes are cached with atomics, nothing else is
    shared: every thread must get its stand-alone results on every iteration, and an error raised in one thread must not leak into another's
    status."""
    import threading
    from v2v_amd import esim, v2e, postops
    u8 = esim.synth_clips(12, 41, 128, 128, dtype=torch.uint8)
    f32 = esim.synth_clips(8, 32, 128, 128, dtype=torch.float32)
    vp = v2e.make_params(24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1)
    jobs = [
        lambda: esim.esim_voxel_batch(u8, [0.2, 0.3, 0.05, 5e-4, 1.0], bin_mode="sum", num_bins=5, seed=3),
        lambda: esim.esim_voxel_batch(f32, [0.2, 0.2, 0.1, 1e-3, 0.1], bin_mode="bilinear", num_bins=5, seed=4),
        lambda: v2e.v2e_voxel_batch(u8, vp, bin_mode="sum", num_bins=5, seed=5),
        lambda: postops.normalize_and_pad(esim.esim_voxel_batch(u8[:, :, :120, :120].contiguous(), [0.2, 0.3, 0.05, 5e-4, 1.0], bin_mode="sum", num_bins=5, seed=6),
                                          normalize=True, method="count"),
    ]
    solo = [j() for j in jobs]
    torch.cuda.synchronize()
    errors, bad_frames = [], torch.zeros((1, 20, 8, 8), dtype=torch.uint8, device="cuda")

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for it in range(25):
                    out = jobs[i]()
                    torch.cuda.current_stream().synchronize()
                    if not torch.equal(out, solo[i]):
                        errors.append(f"thread {i} iteration {it}: result differs from the stand-alone run")
                        return
                    if i == 0 and it % 5 == 0:                      # a refused call in THIS thread: its text stays here
                        try:
                            esim.esim_voxel_batch(bad_frames, [0.2, 0.2, 0, 0, 0], num_bins=5)
                            errors.append("19 pairs into 5 bins was not refused")
                        except AssertionError as e:
                            if "num_bins" not in str(e) and "multiple" not in str(e):
                                errors.append(f"foreign status text: {e}")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
