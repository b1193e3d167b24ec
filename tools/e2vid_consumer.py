"""Load generator for BASELINE config 5: a random-init recurrent UNet with the SHAPE of the reference's E2VIDRecurrent
(model/model.py:216-223 -> model/unet.py:252-310 with the kwargs of config/train_v2v_e2vid_10k.yaml:21-30: 5 bins,
3 encoders, base 32 channels, ConvLSTM, 2 residual blocks, sum skips, bilinear-upsample decoders, 1x1 prediction).
Stock PyTorch-ROCm ops by default -- it is NOT part of the product package (SURVEY: model families are out of scope);
bench.py uses it to measure how fast the fused simulator can feed a consumer.  fused_convlstm=True swaps the three recurrent
blocks for v2v_amd.convlstm.ConvLSTM, the two residual blocks for v2v_amd.convlstm.ResidualBlock and the 5x5 encoder / decoder
convolutions with >= 64 input channels for v2v_amd.convlstm.ConvLayer (the matrix-core kernels of the
recurrent encoder), as do the head, the upsampling and the 1x1 prediction: no stock layer is left on the forward path.
Parameter count matches the reference model: 10,710,401.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _ConvLSTM(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.gates = nn.Conv2d(2 * ch, 4 * ch, 3, padding=1)

    def forward(self, x, state):
        if state is None:
            state = (torch.zeros_like(x), torch.zeros_like(x))
        h, c = state
        i, r, o, g = self.gates(torch.cat([x, h], dim=1)).chunk(4, dim=1)
        c = torch.sigmoid(r) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        return h, (h, c)


class _FusedConvLSTM(nn.Module):
    """The recurrent block on the fused HIP kernel; takes the conv output BEFORE its ReLU.  Parameters live in cell.Gates
    (the reference's name); E2VIDShapedConsumer.load_stock_state_dict maps a stock consumer's `gates` onto them."""

    def __init__(self, ch):
        super().__init__()
        from v2v_amd.convlstm import ConvLSTM
        self.cell = ConvLSTM(ch, ch, 3)

    def forward(self, x, state, input_relu=True):
        h, c = self.cell(x, state, input_relu=input_relu)
        return h, (h, c)


class _Res(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.a, self.b = nn.Conv2d(ch, ch, 3, padding=1), nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return F.relu(self.b(F.relu(self.a(x))) + x)


class _FusedRes(nn.Module):
    """The residual block on the fused matrix-core convolution (v2v_amd.convlstm.ResidualBlock: conv1 / conv2 parameter names)."""

    def __init__(self, ch):
        super().__init__()
        from v2v_amd.convlstm import ResidualBlock
        self.block = ResidualBlock(ch, ch)

    def forward(self, x):
        return self.block(x)


class _FusedConv(nn.Module):
    """A 5x5 encoder (stride 2) or decoder (bilinear x2 upsample, stride 1) convolution + bias + ReLU on the matrix-core kernel
    (v2v_amd.convlstm.ConvLayer: `conv2d` parameter name, as model/submodules.py:ConvLayer / UpsampleConvLayer)."""

    def __init__(self, cin, cout, stride, upsample=False):
        super().__init__()
        from v2v_amd.convlstm import ConvLayer
        self.layer = ConvLayer(cin, cout, 5, stride=stride, padding=2, activation="relu", upsample=upsample)

    def forward(self, x, skip=None):
        return self.layer(x, skip)


class _FusedPred(nn.Module):
    """The 1x1 prediction layer on skip_sum(x, head) in one pass (v2v_amd.convlstm.ConvLayer with kernel_size 1)."""

    def __init__(self, cin):
        super().__init__()
        from v2v_amd.convlstm import ConvLayer
        self.layer = ConvLayer(cin, 1, 1, activation=None)

    def forward(self, x, skip):
        return self.layer(x, skip)


class E2VIDShapedConsumer(nn.Module):
    def __init__(self, num_bins=5, base=32, num_encoders=3, num_res=2, fused_convlstm=False):
        super().__init__()
        self.fused = fused_convlstm
        self.head = _FusedConv(num_bins, base, 1) if fused_convlstm and base == 32 and num_bins <= 8 else nn.Conv2d(num_bins, base, 5, padding=2)
        chans = [base * 2 ** i for i in range(num_encoders + 1)]
        # fused: every 5x5 convolution runs on the matrix-core kernel too (enc1 with its 32 input channels: two taps per K chunk)
        fconv = lambda a, b, s, up=False: _FusedConv(a, b, s, up) if fused_convlstm and (a % 64 == 0 or (a == 32 and b in (64, 128))) else None
        self.enc = nn.ModuleList(fconv(a, b, 2) or nn.Conv2d(a, b, 5, stride=2, padding=2) for a, b in zip(chans[:-1], chans[1:]))
        self.rec = nn.ModuleList((_FusedConvLSTM if fused_convlstm else _ConvLSTM)(b) for b in chans[1:])
        self.res = nn.ModuleList((_FusedRes if fused_convlstm else _Res)(chans[-1]) for _ in range(num_res))
        self.dec = nn.ModuleList(fconv(b, a, 1, True) or nn.Conv2d(b, a, 5, padding=2) for a, b in reversed(list(zip(chans[:-1], chans[1:]))))
        self.pred = _FusedPred(base) if fused_convlstm else nn.Conv2d(base, 1, 1)
        self.states = [None] * num_encoders

    def load_stock_state_dict(self, sd):
        """Load the state_dict of a stock (fused_convlstm=False) consumer into this one, whichever kind it is."""
        if self.fused:
            sd = {k.replace(".gates.", ".cell.Gates.") if k.startswith("rec.") else k: v for k, v in sd.items()}
            sd = {(k.replace(".a.", ".block.conv1.").replace(".b.", ".block.conv2.") if k.startswith("res.") else k): v for k, v in sd.items()}
            fused_convs = {f"{n}.{i}" for n, ml in (("enc", self.enc), ("dec", self.dec)) for i, m in enumerate(ml) if isinstance(m, _FusedConv)}
            fused_convs.add("pred")
            if isinstance(self.head, _FusedConv):
                fused_convs.add("head")
            sd = {(k.rsplit(".", 1)[0] + ".layer.conv2d." + k.rsplit(".", 1)[1] if k.rsplit(".", 1)[0] in fused_convs else k): v for k, v in sd.items()}
        return self.load_state_dict(sd)

    def reset_states(self):
        self.states = [None] * len(self.enc)

    def forward(self, x):
        x = self.head(x) if isinstance(self.head, _FusedConv) else F.relu(self.head(x))        # fused: conv + bias + ReLU in one kernel
        head, blocks = x, []
        for i, (conv, rec) in enumerate(zip(self.enc, self.rec)):
            if isinstance(conv, _FusedConv):
                x, self.states[i] = rec(conv(x), self.states[i], input_relu=False)         # ReLU already applied in the conv's epilogue
            else:
                x, self.states[i] = rec(conv(x) if self.fused else F.relu(conv(x)), self.states[i])
            blocks.append(x)
        for r in self.res:
            x = r(x)
        for i, conv in enumerate(self.dec):
            if isinstance(conv, _FusedConv):
                x = conv(x, blocks[len(blocks) - 1 - i])                               # skip sum + upsample, conv + bias + ReLU
            else:
                x = x + blocks[len(blocks) - 1 - i]                                    # skip_type: sum
                x = F.relu(conv(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)))
        return self.pred(x, head) if isinstance(self.pred, _FusedPred) else self.pred(x + head)


def forward_sequence(model, events, channels_last=False):
    """events [B,T,C,H,W] -> list of T predictions (the time loop of model/train_utils.py:339-345).  channels_last=True feeds
    every time step in torch.channels_last (the model should have been moved with .to(memory_format=torch.channels_last))."""
    model.reset_states()
    if hasattr(model, "forward_sequence"):                                   # the package network: time loop + stream overlap inside
        return model.forward_sequence(events)
    if channels_last and (isinstance(getattr(model, "head", None), _FusedConv) or getattr(model, "reads_any_layout", False)):
        return [model(events[:, t]) for t in range(events.shape[1])]       # the fused head reads any strides (its own layout kernel)
    if channels_last:
        return [model(events[:, t].contiguous(memory_format=torch.channels_last)) for t in range(events.shape[1])]
    return [model(events[:, t]) for t in range(events.shape[1])]


def stock_to_reference_keys(sd):
    """state_dict of a stock E2VIDShapedConsumer -> the reference UNetRecurrent's keys (model/unet.py:252-310), and back with
    reference_to_stock_keys: head.conv2d.*, encoders.N.conv.conv2d.*, encoders.N.recurrent_block.Gates.*, resblocks.N.conv1/2.*,
    decoders.N.conv2d.*, pred.conv2d.*."""
    out = {}
    for k, v in sd.items():
        top, rest = k.split(".", 1)
        if top in ("head", "pred"):
            out[f"{top}.conv2d.{rest}"] = v
        elif top == "enc":
            i, leaf = rest.split(".", 1)
            out[f"encoders.{i}.conv.conv2d.{leaf}"] = v
        elif top == "rec":
            i, _, leaf = rest.split(".", 2)
            out[f"encoders.{i}.recurrent_block.Gates.{leaf}"] = v
        elif top == "res":
            i, ab, leaf = rest.split(".", 2)
            out[f"resblocks.{i}.{'conv1' if ab == 'a' else 'conv2'}.{leaf}"] = v
        elif top == "dec":
            i, leaf = rest.split(".", 1)
            out[f"decoders.{i}.conv2d.{leaf}"] = v
        else:
            raise KeyError(k)
    return out


def reference_to_stock_keys(sd):
    out = {}
    for k, v in sd.items():
        p = k.split(".")
        if p[0] in ("head", "pred"):
            out[f"{p[0]}.{p[-1]}"] = v
        elif p[0] == "encoders" and p[2] == "conv":
            out[f"enc.{p[1]}.{p[-1]}"] = v
        elif p[0] == "encoders":
            out[f"rec.{p[1]}.gates.{p[-1]}"] = v
        elif p[0] == "resblocks":
            out[f"res.{p[1]}.{'a' if p[2] == 'conv1' else 'b'}.{p[-1]}"] = v
        elif p[0] == "decoders":
            out[f"dec.{p[1]}.{p[-1]}"] = v
        else:
            raise KeyError(k)
    return out


if __name__ == "__main__":
    m = E2VIDShapedConsumer()
    print(sum(p.numel() for p in m.parameters()))
