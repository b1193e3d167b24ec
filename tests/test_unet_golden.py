"""Golden G18: the device-kernel consumer (v2v_amd/unet.py, v2v_amd/convlstm.py) against outputs of the REFERENCE'S OWN modules
(model/submodules.py:6-33,68-96,143-235; model/unet.py:252-310) run in float32 on seeded weights by tests/golden/make_goldens.py.

Tolerances are ABSOLUTE and written here.  The kernels compute on bfloat16 operands with float32 accumulation; the reference
ran in float32.  For the values in this fixture (inputs ~N(0,1), activations O(1)):
    single layers   |err| <= 3e-2 max, 6e-3 rms   (one bf16 rounding of inputs + weights, K up to 2304, one bf16 rounding of the output)
    ConvLSTM        |err| <= 1e-2 max on hidden (|h| < 0.4) and cell (float32, never rounded to bf16)
    UNetRecurrent   |err| <= 4e-2 max, 8e-3 rms on the prediction (std 0.78, range -3.9..2.8) at every one of 3 time steps
(the reference's own network under CPU bf16 autocast lands at 1.5e-2 max / 3.4e-3 rms against its float32 self.)
"""
import os

import numpy as np
import pytest

from seeded_weights import load_seeded, seeded_input, seeded_state

gpu = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KW = dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
          num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)


@pytest.fixture(scope="module")
def g18():
    return np.load(os.path.join(HERE, "golden", "g18_unet_modules.npz"))


def _err(got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return float(d.max()), float(np.sqrt((d ** 2).mean()))


# ---- no GPU needed: the package modules carry the reference's state_dict keys and shapes -------------------------------------
def test_package_unet_has_the_reference_state_dict_keys(g18):
    from v2v_amd.unet import E2VIDRecurrent, UNetRecurrent
    net = UNetRecurrent(dict(KW))
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g18["unet__keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g18["unet__shapes"]]
    assert sum(v.numel() for v in sd.values()) == int(g18["unet__n_params"]) == 10710401
    wrapped = E2VIDRecurrent(dict(KW))
    assert list(wrapped.state_dict().keys()) == ["unetrecurrent." + str(k) for k in g18["unet__keys"]]       # model/model.py:203
    assert wrapped.states == [None, None, None]
    wrapped.reset_states()
    for bad in (dict(KW, skip_type="concat"), dict(KW, use_upsample_conv=False), dict(KW, norm="BN"), dict(KW, recurrent_block_type="convgru")):
        with pytest.raises(ValueError):
            UNetRecurrent(bad)


def test_seeded_weight_recipe_reproduces_the_generators_bits(g18):
    shapes = {str(k): tuple(int(x) for x in str(s).split(",")) for k, s in zip(g18["unet__keys"], g18["unet__shapes"])}
    vals = seeded_state(shapes, int(g18["unet__seed"]), float(g18["unet__gain"]))
    probe = np.concatenate([vals[k].ravel()[:3] for k in list(vals)[::5]])
    assert np.array_equal(probe, g18["unet__weight_probe"])


def test_stock_restatement_of_the_network_equals_the_reference_on_cpu(g18):
    """tools/e2vid_consumer.py (the stock-PyTorch network bench.py and the full-size config-5 test use as float32 yardstick) is a
    restatement: this pins it to the reference's UNetRecurrent outputs -- float32 on the CPU, same seeded weights through the key
    map, 3 recurrent steps; float32 summation order is the only freedom (1e-4 absolute on values up to 3.9)."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from e2vid_consumer import E2VIDShapedConsumer, forward_sequence, reference_to_stock_keys, stock_to_reference_keys
    torch.set_num_threads(4)
    net = E2VIDShapedConsumer(num_bins=5).eval()
    shapes = {str(k): tuple(int(x) for x in str(s).split(",")) for k, s in zip(g18["unet__keys"], g18["unet__shapes"])}
    vals = seeded_state(shapes, int(g18["unet__seed"]), float(g18["unet__gain"]))
    sd = reference_to_stock_keys({k: torch.from_numpy(v) for k, v in vals.items()})
    assert set(stock_to_reference_keys(sd)) == set(vals)
    net.load_state_dict(sd, strict=True)
    vox = torch.from_numpy(g18["unet__vox"].astype(np.float32)).permute(1, 0, 2, 3, 4)          # [B=2, T=3, 5, 64, 64]
    with torch.no_grad():
        imgs = forward_sequence(net, vox)
    for t in range(3):
        mx, _ = _err(imgs[t].numpy(), g18["unet__images"][t])
        assert mx <= 1e-4, (t, mx)


# ---- GPU: device kernels vs the reference's outputs ---------------------------------------------------------------------------
@gpu
def test_convlstm_two_steps_vs_reference(g18):
    import torch
    from v2v_amd.convlstm import ConvLSTM
    m = ConvLSTM(64, 64, 3).cuda().eval()
    load_seeded(m, int(g18["convlstm__seed"]))
    with torch.no_grad():
        x0, x1 = (torch.from_numpy(seeded_input(sd, *g18["convlstm__x_shape"])).cuda() for sd in g18["convlstm__x_seeds"])
        h1, c1 = m(x0, None)
        h2, c2 = m(x1, (h1, c1))
    for name, got in (("h1", h1), ("c1", c1), ("h2", h2), ("c2", c2)):
        mx, rms = _err(got.float().cpu().numpy(), g18["convlstm__" + name])
        assert mx <= 1e-2 and rms <= 2e-3, (name, mx, rms)


@gpu
@pytest.mark.parametrize("layout", ["nchw_f32", "channels_last_bf16"])
def test_single_layers_vs_reference(g18, layout):
    import torch
    from v2v_amd.convlstm import ConvLayer, ResidualBlock
    from v2v_amd.unet import UpsampleConvLayer
    cases = [("resblock", ResidualBlock(256, 256)), ("convlayer", ConvLayer(64, 128, 5, stride=2, padding=2)),
             ("upsample", UpsampleConvLayer(128, 64, 5, padding=2))]
    for name, mod in cases:
        mod = mod.cuda().eval()
        load_seeded(mod, int(g18[name + "__seed"]))
        x = torch.from_numpy(seeded_input(g18[name + "__x_seed"], *g18[name + "__x_shape"])).cuda()
        if layout == "channels_last_bf16":
            x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = mod(x)
        assert tuple(y.shape) == g18[name + "__y"].shape
        mx, rms = _err(y.float().cpu().numpy(), g18[name + "__y"])
        assert mx <= 3e-2 and rms <= 6e-3, (name, layout, mx, rms)


@gpu
def test_unet_recurrent_three_steps_vs_reference(g18):
    """UNetRecurrent(num_bins 5, base 32, 3 encoders, 2 residual blocks, sum skips) loaded through the REFERENCE'S state_dict keys,
    3 time steps at 64x64 with the recurrent state carried, float32 NCHW voxel grids in, float32 prediction out."""
    import torch
    from v2v_amd.unet import E2VIDRecurrent
    net = E2VIDRecurrent(dict(KW)).cuda().eval()
    shapes = {str(k): tuple(int(x) for x in str(s).split(",")) for k, s in zip(g18["unet__keys"], g18["unet__shapes"])}
    vals = seeded_state(shapes, int(g18["unet__seed"]), float(g18["unet__gain"]))
    missing = net.load_state_dict({"unetrecurrent." + k: torch.from_numpy(v) for k, v in vals.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    vox = torch.from_numpy(g18["unet__vox"].astype(np.float32)).cuda()
    net.reset_states()
    with torch.no_grad():
        for t in range(3):
            img = net(vox[t])["image"]
            assert img.dtype == torch.float32 and tuple(img.shape) == (2, 1, 64, 64)
            mx, rms = _err(img.cpu().numpy(), g18["unet__images"][t])
            assert mx <= 4e-2 and rms <= 8e-3, (t, mx, rms)
    st = net.states                                                   # model/model.py:205-207: a copy, (hidden, cell) per encoder
    assert len(st) == 3 and all(isinstance(s, tuple) and len(s) == 2 for s in st)
    for s, want in zip(st, g18["unet__hidden0_absmax"]):
        assert abs(float(s[0].float().abs().max()) - float(want)) <= 2e-2
    # the same network fed channels-last bfloat16 under autocast (how bench.py's config 5 runs it) gives the same prediction
    net.reset_states()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for t in range(3):
            img = net(vox[t].contiguous(memory_format=torch.channels_last))["image"]
            mx, rms = _err(img.float().cpu().numpy(), g18["unet__images"][t])
            assert img.dtype == torch.bfloat16 and mx <= 6e-2 and rms <= 1.2e-2, (t, mx, rms)     # + one bf16 rounding of the output (|img| < 4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 7, 64, 64), (12, 6, 128, 128)])
def test_forward_sequence_pipelined_over_two_streams_equals_the_step_loop(shape):
    """E2VIDRecurrent.forward_sequence (the reference's time loop, model/train_utils.py:339-345, in one call): the decoder half of step t
    on a second stream under the encoder half of step t + 1 gives bit-identical images and states to calling the network step by step --
    eagerly and replayed from a captured hipGraph (fork / join through events).  (Round 5: the first version of this overlap exposed wrong
    packed-float32 results under co-scheduling; the library carries no such instructions any more, tests/test_kernel_resources.py.)"""
    import torch
    from v2v_amd.unet import E2VIDRecurrent
    n, t, h, w = shape
    torch.manual_seed(1)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
    ev = torch.round(torch.randn((n, t, 5, h, w), device="cuda") * 2)
    sc = torch.full((n, 2), 3.0, device="cuda")
    with torch.no_grad():
        net.reset_states()
        want = torch.stack([net(ev[:, i], sc)["image"] for i in range(t)], 1)
        want_states = net.states
        for rep in range(3):                                               # a race shows up as run-to-run differences
            net.reset_states()
            got = net.forward_sequence(ev, sc)
            assert got.shape == (n, t, 1, h, w) and got.dtype == ev.dtype and torch.equal(got, want), rep
            for a, b in zip(net.states, want_states):
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for ov in (False, 1, 2):                                          # the plain loop; one / two side streams (True = three)
            net.reset_states()
            assert torch.equal(net.forward_sequence(ev, sc, overlap=ov), want), ov
        # two calls continue the recurrence: states carry over
        net.reset_states()
        first = net.forward_sequence(ev[:, :3], sc)
        second = net.forward_sequence(ev[:, 3:], sc)
        assert torch.equal(torch.cat([first, second], 1), want)
        # the built-in graph cache: captured on the first call, replayed afterwards, new inputs flow through the static buffers
        ev2 = torch.round(torch.randn((n, t, 5, h, w), device="cuda") * 2)
        net.reset_states()
        want2 = torch.stack([net(ev2[:, i], sc)["image"] for i in range(t)], 1)
        want2_states = net.states
        for rep in range(2):
            assert torch.equal(net.forward_sequence(ev, sc, graph=True), want), rep
            got2 = net.forward_sequence(ev2, sc, graph=True)
            assert torch.equal(got2, want2), rep
            for a, b in zip(net.states, want2_states):
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert len(net._sequence_graphs) == 1
        # captured into one hipGraph by hand
        out = torch.empty_like(want)

        def run():
            net.reset_states()
            net.forward_sequence(ev, sc, out=out)
        run()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run()
        for rep in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want), rep


@pytest.mark.gpu
def test_forward_sequence_as_the_first_call_and_after_a_weight_update():
    """the layers pack their weights lazily inside forward(); when forward_sequence is the FIRST call on a network (or
    follows load_state_dict / a weight update) the decoder halves would pack on side stream 0 while side streams 1 and 2 read the packed
    copies ordered only after the caller's event.  forward_sequence now packs everything on the caller's stream before the fork: a fresh
    network's first overlapped call equals the step loop of an identical twin, repeatedly, and again after the weights change in place.
    Also (low): an eager call between two graph replays must not leak its states into the next replay's result."""
    import torch
    from v2v_amd.unet import E2VIDRecurrent
    kw = dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)
    n, t, h, w = 4, 6, 64, 64
    ev = torch.round(torch.randn((n, t, 5, h, w), device="cuda", generator=torch.Generator("cuda").manual_seed(3)) * 2)
    with torch.no_grad():
        for trial in range(4):
            torch.manual_seed(10 + trial)
            fresh = E2VIDRecurrent(kw).cuda().eval()
            twin = E2VIDRecurrent(kw).cuda().eval()
            twin.load_state_dict(fresh.state_dict())
            got = fresh.forward_sequence(ev)                                   # FIRST call of this network: packs, then forks
            twin.reset_states()
            want = torch.stack([twin(ev[:, i])["image"] for i in range(t)], 1)
            assert torch.equal(got, want), trial
            for p, q in zip(fresh.parameters(), twin.parameters()):            # in-place update: every packed copy is stale now
                p.mul_(0.5)
                q.mul_(0.5)
            fresh.reset_states()
            twin.reset_states()
            got = fresh.forward_sequence(ev)
            want = torch.stack([twin(ev[:, i])["image"] for i in range(t)], 1)
            assert torch.equal(got, want), trial
        # graph replays with an eager call in between
        net = fresh
        net.reset_states()
        want_a = torch.stack([net(ev[:, i])["image"] for i in range(t)], 1)
        states_a = net.states
        assert torch.equal(net.forward_sequence(ev, graph=True), want_a)
        net(ev[:, 0])                                                          # eager step: writes into the live state list
        assert torch.equal(net.forward_sequence(ev, graph=True), want_a)       # the replay resets and recomputes
        for a, b in zip(net.states, states_a):                                 # and hands back ITS final states, not the eager call's
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
