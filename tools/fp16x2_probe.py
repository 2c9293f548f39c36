"""Representation error of operand-splitting schemes for fp32 GEMMs on 16-bit matrix pipes (CPU, float64 arithmetic on
the pieces, so only the splitting is measured): bf16 x 3 pieces / 6 products (what FCD_PREC_BF16X3 runs), fp16 x 2
pieces / 3 or 4 products with a per-tensor power-of-two scale, plain fp32 rounding of the products for scale."""
import torch
torch.manual_seed(0)


def split(x, dtype, n):
    pieces, r = [], x.clone()
    for _ in range(n):
        p = r.to(dtype).to(torch.float32)
        pieces.append(p)
        r = r - p
    return pieces


def scale_pow2(x, target=2.0 ** 13):
    return 2.0 ** torch.floor(torch.log2(target / x.abs().max()))


def run(name, a, b):
    ref = a.double() @ b.double()
    nrm = ref.norm()
    out = {}
    out["fp32 operands, fp32 accumulate"] = ((a @ b).double() - ref).norm() / nrm
    ah, am, al = split(a, torch.bfloat16, 3); bh, bm, bl = split(b, torch.bfloat16, 3)
    d = lambda t: t.double()
    six = d(ah) @ d(bh) + d(ah) @ d(bm) + d(am) @ d(bh) + d(ah) @ d(bl) + d(al) @ d(bh) + d(am) @ d(bm)
    out["bf16 x3, 6 products (exact sums)"] = (six - ref).norm() / nrm
    sa, sb = scale_pow2(a), scale_pow2(b)
    ah, al = split(a * sa, torch.float16, 2); bh, bl = split(b * sb, torch.float16, 2)
    three = (d(ah) @ d(bh) + d(ah) @ d(bl) + d(al) @ d(bh)) / (sa * sb).double()
    four = three + d(al) @ d(bl) / (sa * sb).double()
    out["fp16 x2 (per-tensor scale), 3 products"] = (three - ref).norm() / nrm
    out["fp16 x2 (per-tensor scale), 4 products"] = (four - ref).norm() / nrm
    # worst output ROW (relative to that row's norm): what a channel with small values sees
    rown = ref.norm(dim=1)
    out["  worst row, bf16 x3"] = ((six - ref).norm(dim=1) / rown).max()
    out["  worst row, fp16 x2 / 3 products"] = ((three - ref).norm(dim=1) / rown).max()
    print(name)
    for k, v in out.items():
        print(f"   {k:42s} {float(v):.2e}")


M, K, N = 512, 2304, 256
a = torch.randn(M, K); b = torch.randn(K, N) / 48
run("gaussian operands", a, b)
rs = torch.exp(torch.randn(M, 1) * 4.0)            # rows whose scales differ by e^+-12
run("rows of A scaled by exp(N(0, 4^2)) (channels of very different magnitude)", a * rs, b)
run("tiny gradients (1e-9 scale)", a * 1e-9, b)
relu = torch.relu(torch.randn(M, K)) * torch.exp(torch.randn(1, K))
run("post-ReLU activations with per-channel scales", relu, b)
