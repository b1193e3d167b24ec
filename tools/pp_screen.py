"""Race screen for an experimental tile of the fused ConvLSTM step: the same per-accumulator summation order as the shipped tiles,
so the outputs must be bit-identical to theirs -- many launches, several shapes, with and without state, under memory load."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import convlstm as CL  # noqa: E402


def main(exp=257, ref=256, rounds=30):
    bad = 0
    noise = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    for (b, c, h, w) in [(8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32), (2, 64, 16, 16), (1, 128, 16, 16), (3, 64, 32, 32), (1, 256, 16, 16)]:
        torch.manual_seed(b * c + h)
        weight = torch.randn((4 * c, 2 * c, 3, 3), device="cuda") * (1.0 / (18 * c) ** 0.5)
        bias = torch.randn((4 * c,), device="cuda") * 0.1
        packed = CL.pack_gate_weights(weight)
        for r in range(rounds):
            x = torch.randn((b, h, w, c), device="cuda").to(torch.bfloat16)
            hp = torch.randn((b, h, w, c), device="cuda").to(torch.bfloat16) if r % 3 else None
            cp = torch.randn((b, h, w, c), device="cuda") if hp is not None else None
            want = CL.convlstm_step(x, hp, cp, packed, bias, tile_rows=ref)
            if r % 2:
                with torch.cuda.stream(side):                 # memory traffic beside the launch: moves the DMA timing
                    noise.add_(1)
            got = CL.convlstm_step(x, hp, cp, packed, bias, tile_rows=exp)
            torch.cuda.synchronize()
            for k, (g, wv) in enumerate(zip(got, want)):
                if g is not None and not torch.equal(g, wv):
                    bad += 1
                    d = (g.float() - wv.float()).abs()
                    print("MISMATCH", (b, c, h, w), "round", r, "output", k, "max", float(d.max()), "count", int((d > 0).sum()), flush=True)
        print("shape", (b, c, h, w), "done, mismatches so far", bad, flush=True)
    print("RESULT", "ok" if bad == 0 else f"{bad} mismatching outputs")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(v) for v in sys.argv[1:])) else 0)
