"""The recurrent UNet that consumes the voxel grids (BASELINE config 5), assembled from the device kernels of
v2v_amd/convlstm.py under the REFERENCE'S OWN module tree, so that a reference checkpoint loads unchanged:

    E2VIDRecurrent(unet_kwargs)            model/model.py:194-223          keys  unetrecurrent.*
    UNetRecurrent(unet_kwargs)             model/unet.py:252-310           keys  head.conv2d.*, encoders.N.conv.conv2d.*,
                                                                                 encoders.N.recurrent_block.Gates.*,
                                                                                 resblocks.N.conv1.* / conv2.*,
                                                                                 decoders.N.conv2d.*, pred.conv2d.*
    RecurrentConvLayer / UpsampleConvLayer model/submodules.py:99-119 / :68-96

Configuration covered = what config/train_v2v_e2vid_10k.yaml:21-30 instantiates: skip_type 'sum', recurrent_block_type
'convlstm', use_upsample_conv true, norm none, kernel_size 5, base_num_channels 32, channel_multiplier 2 (anything else raises:
there is no stock-layer fallback inside this module).  Inference only (the kernels have no backward).

Inside forward() everything runs in bfloat16 NHWC (torch.channels_last views of the kernels' own buffers): the head takes
the float voxel grid in any layout, every later layer consumes and produces NHWC in place, the cell states stay float32, and
the prediction comes back in the input's dtype.  Parity: golden G18 = the reference's modules run in float32 on seeded
weights (tests/test_unet_golden.py; tolerance stated there).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .convlstm import ConvLayer, ConvLSTM, ResidualBlock


class UpsampleConvLayer(ConvLayer):
    """model/submodules.py:68-96: bilinear x2 upsampling + convolution + ReLU; forward(x, skip) == forward(skip_sum(x, skip))."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation="relu", norm=None):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, activation=activation, norm=norm,
                         upsample=True)


class RecurrentConvLayer(nn.Module):
    """model/submodules.py:99-119: ConvLayer followed by the ConvLSTM; same attribute names (`conv`, `recurrent_block`)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, recurrent_block_type="convlstm",
                 activation="relu", norm=None, BN_momentum=0.1):
        super().__init__()
        if recurrent_block_type != "convlstm":
            raise ValueError("only recurrent_block_type 'convlstm' runs on the device kernels")
        self.recurrent_block_type = recurrent_block_type
        self.conv = ConvLayer(in_channels, out_channels, kernel_size, stride, padding, activation, norm)
        self.recurrent_block = ConvLSTM(input_size=out_channels, hidden_size=out_channels, kernel_size=3)

    def forward(self, x, prev_state, conv_out=None):
        """conv_out (this implementation only): self.conv(x) when the caller has it already -- the convolution does not touch the state,
        so UNetRecurrent.forward_sequence runs the first encoder's for all time steps in one launch."""
        x = self.conv(x) if conv_out is None else conv_out  # ReLU in the convolution's epilogue
        state = self.recurrent_block(x, prev_state)
        return state[0], state


class UNetRecurrent(nn.Module):
    """model/unet.py:252-310 (+ BaseUNet :13-64).  `unet_kwargs` as the reference's YAML gives them."""

    def __init__(self, unet_kwargs):
        super().__init__()
        kw = dict(unet_kwargs)
        final_activation = kw.pop("final_activation", "none")
        self.final_activation = getattr(torch, final_activation, None) if final_activation else None
        kw["num_output_channels"] = 1                      # :263
        self.base_num_channels = kw["base_num_channels"]
        self.num_encoders = kw["num_encoders"]
        self.num_residual_blocks = kw["num_residual_blocks"]
        self.num_output_channels = kw["num_output_channels"]
        self.kernel_size = kw.get("kernel_size", 5)
        self.skip_type = kw["skip_type"]
        self.norm = kw.get("norm", None)
        self.num_bins = kw["num_bins"]
        self.recurrent_block_type = kw.get("recurrent_block_type", None)
        mult = kw.get("channel_multiplier", 2)
        if self.norm in ("none", "None", ""):
            self.norm = None
        if self.skip_type != "sum" or not kw.get("use_upsample_conv", True) or self.norm is not None:
            raise ValueError("the device kernels cover skip_type 'sum', use_upsample_conv true, norm none "
                             "(config/train_v2v_e2vid_10k.yaml:21-30)")
        self.encoder_input_sizes = [int(self.base_num_channels * pow(mult, i)) for i in range(self.num_encoders)]
        self.encoder_output_sizes = [int(self.base_num_channels * pow(mult, i + 1)) for i in range(self.num_encoders)]
        self.max_num_channels = self.encoder_output_sizes[-1]
        k = self.kernel_size
        self.head = ConvLayer(self.num_bins, self.base_num_channels, kernel_size=k, stride=1, padding=k // 2)
        self.head.force_channels_last = True               # NHWC from the first layer on, whatever layout the voxel grid arrives in
        self.encoders = nn.ModuleList(
            RecurrentConvLayer(i, o, kernel_size=k, stride=2, padding=k // 2, recurrent_block_type=self.recurrent_block_type, norm=self.norm)
            for i, o in zip(self.encoder_input_sizes, self.encoder_output_sizes))
        self.resblocks = nn.ModuleList(ResidualBlock(self.max_num_channels, self.max_num_channels, norm=self.norm)
                                       for _ in range(self.num_residual_blocks))
        self.decoders = nn.ModuleList(UpsampleConvLayer(i, o, kernel_size=k, padding=k // 2, norm=self.norm)
                                      for i, o in zip(reversed(self.encoder_output_sizes), reversed(self.encoder_input_sizes)))
        self.pred = ConvLayer(self.base_num_channels, self.num_output_channels, 1, activation=None, norm=self.norm)
        self.states = [None] * self.num_encoders

    def _encode(self, x, event_scales, head=None, conv0=None):
        """head + the recurrent encoders of one time step (model/unet.py:287-296) -> (head, blocks): everything that touches the states.
        `head` / `conv0`: the head layer's and the first encoder convolution's outputs for this step when the caller computed them
        already (forward_sequence does, for all steps at once: neither depends on the states)."""
        if head is None:
            with torch.autocast("cuda", dtype=torch.bfloat16):  # the head hands out bfloat16; every later layer keeps it
                head = self.head(x, scales=event_scales)        # reads any strides (its own layout kernel), bfloat16 NHWC out
        x = head
        blocks = []
        for i, encoder in enumerate(self.encoders):
            x, state = encoder(x, self.states[i], conv_out=conv0 if i == 0 else None)
            blocks.append(x)
            self.states[i] = state
        return head, blocks

    def _decode(self, head, blocks):
        """residual blocks + decoders + prediction of one time step (model/unet.py:298-309): stateless."""
        x = blocks[-1]
        for resblock in self.resblocks:
            x = resblock(x)
        for i, decoder in enumerate(self.decoders):
            x = decoder(x, blocks[self.num_encoders - i - 1])   # skip_sum folded into the upsampling kernel (:304)
        img = self.pred(x, head)                                 # pred(skip_sum(x, head)) in one pass (:307)
        if self.final_activation is not None:
            img = self.final_activation(img)
        return img

    def _pack_weights(self):
        """Every layer's packed weight copy made (or found current) on the CALLER's stream.  The layers pack lazily inside forward(); a first
        call, or one after load_state_dict / a weight update, would otherwise pack the decoder halves' weights on the side stream of step
        0 -- and step 1's decoder half, on ANOTHER side stream ordered only after the caller's `ready` event, could read them before the
        pack kernels have written them (and the packed tensors would live in one side stream's allocator pool while every stream reads
        them).  Packed here, they are ordered before every side stream by the wait_stream() that follows."""
        for m in self.modules():
            if isinstance(m, ResidualBlock):
                m._weights(m.conv1, "conv1")
                m._weights(m.conv2, "conv2")
            elif isinstance(m, ConvLSTM) or (isinstance(m, ConvLayer) and m.conv2d.kernel_size[0] != 1):   # the 1x1 prediction layer has no packed copy
                m._weights()

    def forward_sequence(self, events, event_scales=None, out=None, overlap=True):
        """The time loop of model/train_utils.py:339-345 (`for t in range(T): pred = model(events[:, t]); pred_imgs[:, t] = pred['image']`)
        as ONE call: events [N,T,num_bins,H,W] -> images [N,T,1,H,W] (events' dtype, or `out`), the states advanced by T steps.

        The recurrence only runs through the encoders' ConvLSTM states; residual blocks, decoders and prediction of step t are stateless.
        With overlap they are issued on side HIP streams (overlap=True: three, taken in turn by consecutive steps; an int: that many),
        so step t's decoder half runs UNDER step t+1's encoder half and beside its neighbours' decoder halves: at the training shape
        (12 x 128 x 128) no single layer fills 256 CUs (48-384 workgroups), and half-filling kernels side by side use what one leaves
        idle (ms per time step, hipGraph replay, tools/e2vid_pipeline_probe.py: loop 0.50, one side stream 0.37, two 0.355, three 0.34;
        0.316 with the state-free layers batched over time, below).  Same kernels on the same operands in the same per-tensor order:
        results are bit-identical to the step-by-step loop (tests/test_unet_golden.py).
        Stream-ordering contract with torch's caching allocator: tensors made on the caller's stream and read on the side stream
        (head, skip blocks) are kept alive until the caller's stream has waited for the side stream's event of that step, so a freed
        block can never be handed out again while the side stream still reads it.  Captures into a hipGraph (fork / join through
        events) like the single-stream loop.  (A three-STAGE form -- the decoder half itself split over two chained side streams -- ran
        eagerly but crashed hipGraph's capture_end on ROCm 7.2; whole decoder halves on alternating streams capture fine.)"""
        if events.dim() != 5:
            raise ValueError("events must be [N, T, num_bins, H, W]")
        n, t_steps = events.shape[:2]
        out_dtype = torch.bfloat16 if (events.dtype == torch.bfloat16 or torch.is_autocast_enabled()) else events.dtype
        if out is None:
            out = torch.empty((n, t_steps, self.num_output_channels) + tuple(events.shape[-2:]), dtype=out_dtype, device=events.device)
        if not overlap:
            for t in range(t_steps):
                head, blocks = self._encode(events[:, t], event_scales)
                out[:, t] = self._decode(head, blocks)
            return out
        # The head layer and the first encoder's convolution do not touch the states: all T steps' in ONE well-filled launch each (T*N
        # images) instead of T small ones on the critical path of the recurrence (0.339 -> 0.322 -> 0.30 ms per step at the training
        # shape).  Per image the kernels do the same work whatever the batch is, so the values are those of the per-step calls bit for
        # bit.  Capped at 2 GiB of bf16 activations (heads: 32 channels at full resolution; conv0: 64 at half).
        heads = conv0s = None
        c_head = self.base_num_channels
        if t_steps > 1 and n * t_steps * events.shape[-2] * events.shape[-1] * c_head * 3 <= (2 << 30):
            ev_t = events.transpose(0, 1).reshape((t_steps * n,) + tuple(events.shape[2:]))          # t-major copy: step t = rows t*N .. (t+1)*N
            sc_t = event_scales.repeat(t_steps, 1) if event_scales is not None else None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                heads = self.head(ev_t, scales=sc_t)
                if n * events.shape[-2] * events.shape[-1] <= 16 * 128 * 128:   # larger steps fill the chip themselves (8 x 256^2: batching the
                    conv0s = self.encoders[0].conv(heads)                      # convolution measured 0.690 -> 0.70 ms per step, it stays per step)
            del ev_t, sc_t
        cur = torch.cuda.current_stream(events.device)
        n_side = 3 if overlap is True else max(1, int(overlap))  # decoder halves of consecutive steps alternate between the side streams
        pool = self.__dict__.setdefault("_side_streams", {})
        key = (events.device, n_side)
        if key not in pool:
            pool[key] = [torch.cuda.Stream(device=events.device) for _ in range(n_side)]
        sides = pool[key]
        self._pack_weights()                                    # on `cur`, BEFORE the fork: see _pack_weights
        for side in sides:
            side.wait_stream(cur)                               # `out`, the weights' packed copies, whatever the caller queued before
        held = []                                               # (tensors of a step a side stream reads, its completion event)
        for t in range(t_steps):
            if len(held) == n_side + 1:                         # the oldest step's operands may go once cur is ordered after their last reader
                cur.wait_event(held[0][1])
                held.pop(0)
            head, blocks = self._encode(events[:, t], event_scales, head=None if heads is None else heads[t * n:(t + 1) * n],
                                        conv0=None if conv0s is None else conv0s[t * n:(t + 1) * n])
            ready = torch.cuda.Event()
            ready.record(cur)
            side = sides[t % n_side]
            with torch.cuda.stream(side):
                side.wait_event(ready)
                out[:, t] = self._decode(head, blocks)
                done = torch.cuda.Event()
                done.record(side)
            held.append(((head, blocks), done))
        for side in sides:
            cur.wait_stream(side)
        return out

    def forward(self, x, event_scales=None):
        """x: [N, num_bins, H, W] float voxel grid (any layout), H and W multiples of 16 (what forward_sequence pads to, model/train_utils.py:322-326; anything else raises) -> {'image': [N,1,H,W]}.
        event_scales (this implementation only): float32 [N,2] = (neg_max, pos_max) per sample, e.g. RingLoader(normalize='scales')'s
        batch['event_scales'] -- normalize_batch_voxel (model/train_utils.py:147-166) is then applied by the head while it reads the
        RAW voxel grid; None = x is used as it is."""
        out_dtype = torch.bfloat16 if (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()) else x.dtype
        head, blocks = self._encode(x, event_scales)
        return {"image": self._decode(head, blocks).to(out_dtype)}


def copy_states(states):
    """model/model.py:17-24 copy_states: clone every state tensor (a list of None stays a list of None)."""
    if states[0] is None:
        return list(states)
    return [tuple(s.detach().clone() for s in st) if isinstance(st, tuple) else st.detach().clone() for st in states]


class E2VIDRecurrent(nn.Module):
    """model/model.py:194-223: `unetrecurrent` + the states property / reset_states the training loop uses."""

    def __init__(self, unet_kwargs):
        super().__init__()
        self.num_bins = unet_kwargs["num_bins"]
        self.num_encoders = unet_kwargs["num_encoders"]
        self.unetrecurrent = UNetRecurrent(unet_kwargs)

    @property
    def states(self):
        return copy_states(self.unetrecurrent.states)

    @states.setter
    def states(self, states):
        self.unetrecurrent.states = states

    def reset_states(self):
        self.unetrecurrent.states = [None] * self.unetrecurrent.num_encoders

    def forward(self, event_tensor, event_scales=None):
        return self.unetrecurrent.forward(event_tensor, event_scales)

    def forward_sequence(self, events, event_scales=None, out=None, overlap=True, graph=False):
        """[N,T,num_bins,H,W] -> [N,T,1,H,W]: the reference's time loop (model/train_utils.py:339-345) in one call, decoder half of step t
        under the encoder half of step t+1 (UNetRecurrent.forward_sequence).

        graph=True: the whole sequence -- reset_states() first, as forward_sequence(reset_states=True) does (model/train_utils.py:309-313),
        then T steps on the overlapped streams -- is captured ONCE per (shape, dtype, device, weights) into a hipGraph and replayed from
        then on: ~20 launches per time step cost the host ~12 ms per 40-step sequence when issued one by one, about what the GPU needs to
        run them.  Inputs are copied into the graph's static buffers; the returned tensor is the graph's static output (overwritten by
        the next call with the same shapes -- clone it to keep it); the states after the call are those of the sequence's last step.
        Inference only, like every layer here."""
        if not graph:
            return self.unetrecurrent.forward_sequence(events, event_scales, out=out, overlap=overlap)
        if out is not None:
            raise ValueError("graph=True returns the captured graph's own output buffer; `out` is not supported")
        params = list(self.parameters())
        key = (tuple(events.shape), events.dtype, events.device, event_scales is not None, overlap,
               tuple(p.data_ptr() for p in params), sum(p._version for p in params))
        cache = self.__dict__.setdefault("_sequence_graphs", {})
        entry = cache.get(key)
        if entry is None:
            cache.clear()                                       # one live graph: a new shape or new weights retire the old one and its buffers
            ev = torch.empty_like(events, memory_format=torch.contiguous_format)
            sc = torch.empty_like(event_scales, memory_format=torch.contiguous_format) if event_scales is not None else None

            def run():
                self.reset_states()
                return self.unetrecurrent.forward_sequence(ev, sc, overlap=overlap)
            ev.copy_(events)
            if sc is not None:
                sc.copy_(event_scales)
            with torch.no_grad():
                run()                                           # eager once: weight packing, LDS-size attributes, allocator warm-up
                torch.cuda.synchronize(events.device)
                warm = torch.cuda.Stream(device=events.device)
                warm.wait_stream(torch.cuda.current_stream(events.device))
                with torch.cuda.stream(warm):
                    run()
                torch.cuda.current_stream(events.device).wait_stream(warm)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    res = run()
            entry = cache[key] = (g, ev, sc, res, tuple(self.unetrecurrent.states))   # an immutable copy: eager calls between two replays
        g, ev, sc, res, states = entry                                                # assign into the live list (self.states[i] = state)
        ev.copy_(events)
        if sc is not None:
            sc.copy_(event_scales)
        g.replay()
        self.unetrecurrent.states = list(states)                # the graph's own state tensors: what the last captured step wrote
        return res
