"""Would F(4x4, 3x3) Winograd in fp32 pass the conv parity gates?  CPU emulation (torch, float32 arithmetic in every stage, fp32
accumulation over channels in the kernel's chunk order) of F(2x2,3x3) -- the shipped kernels' algorithm -- and F(4x4,3x3) against a
float64 convolution, on layer shapes of the bench model.  Error measure = the conv tests' (max |y - ref| / max |ref|).
    python tools/wino43_error.py
"""
import torch, torch.nn.functional as F

torch.manual_seed(0)
f32, f64 = torch.float32, torch.float64

# F(2,3)
B2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=f64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=f64)
A2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=f64)
# F(4,3) (Lavin & Gray), interpolation points 0, +-1, +-2, inf
B4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=f64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=f64)
A4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=f64)


def wino(x, w, BT, G, AT, m, chunk=8):
    """x [N,C,H,W] (H, W multiples of m), w [M,C,3,3]; zero padding 1; every stage in fp32."""
    N, C, H, W = x.shape
    M = w.shape[0]
    a = m + 2
    BT, G, AT = BT.to(f32), G.to(f32), AT.to(f32)
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                      # [N,C,th,tw,a,a]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)          # B^T d B   (fp32)
    U = torch.einsum("ij,mcjk,lk->mcil", G, w, G)                    # G g G^T
    acc = torch.zeros(N, M, V.shape[2], V.shape[3], a, a, dtype=f32)
    for c0 in range(0, C, chunk):                                    # fp32 accumulation, chunk by chunk like the kernel's K loop
        acc = acc + torch.einsum("mcil,nctuil->nmtuil", U[:, c0:c0 + chunk], V[:, c0:c0 + chunk])
    Y = torch.einsum("ij,nmtujk,lk->nmtuil", AT, acc, AT)            # A^T (.) A
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, M, H, W)


def rel(y, r):
    return float((y.double() - r).abs().max() / r.abs().max())


print("shape (N, Cin, H, W, Cout)         direct fp32    F(2x2,3x3)    F(4x4,3x3)    ratio 4/2")
for (N, C, H, W, M, kind) in [(1, 64, 32, 32, 64, "randn"), (1, 256, 16, 32, 256, "randn"), (1, 512, 16, 32, 512, "randn"),
                              (1, 256, 16, 32, 256, "relu"), (1, 512, 16, 32, 128, "relu"), (1, 128, 32, 64, 128, "offset")]:
    x = torch.randn(N, C, H, W)
    if kind == "relu":
        x = F.relu(x)                                               # post-activation inputs (VGG / generator): non-zero mean
    if kind == "offset":
        x = x + 3.0
    w = torch.randn(M, C, 3, 3) / (C * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    e0 = rel(F.conv2d(x, w, padding=1), ref)
    e2 = rel(wino(x, w, B2, G2, A2, 2), ref)
    e4 = rel(wino(x, w, B4, G4, A4, 4), ref)
    print(f"{str((N, C, H, W, M)):28s} {kind:7s} {e0:10.2e}    {e2:10.2e}    {e4:10.2e}    {e4 / e2:6.1f}")
print("conv parity gates of tests/test_gpu_ops.py: 2e-5 (forward), 5e-5 (data gradient), 1e-4 (weight gradient)")
