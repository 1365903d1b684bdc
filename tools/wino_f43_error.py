#!/usr/bin/env python3
"""How much accuracy would Winograd F(4x4,3x3) cost in fp32?  (CPU, torch)
Evaluates a 3x3 convolution with F(2x2,3x3) - what conv_wino_kernel computes - and with
F(4x4,3x3) (interpolation points 0, +-1, +-2, inf), all transforms and the element-wise products
in fp32, against the direct convolution in fp64.  Prints max and rms error relative to max |y|.
The FLOP ratio of the multiply stage: direct 36 / F(2,3) 16 / F(4,3) 9 per 2x2 outputs."""
import torch
torch.manual_seed(0)

def mats(kind, dt):
    if kind == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=dt)
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=dt)
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=dt)
    else:
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                           [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=dt)
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                          [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=dt)
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0],
                           [0, 1, -1, 8, -8, 1]], dtype=dt)
    return BT, G, AT

def wino(x, w, kind, dt=torch.float32):
    """x [N,C,H,W] (H, W multiples of the output tile), w [K,C,3,3]; zero padding 1."""
    m = 2 if kind == 2 else 4
    t = m + 2
    BT, G, AT = mats(kind, dt)
    x = torch.nn.functional.pad(x.to(dt), (1, 1, 1, 1))
    N, C, H, W = x.shape
    th, tw = (H - 2) // m, (W - 2) // m
    tiles = x.unfold(2, t, m).unfold(3, t, m)                 # [N,C,th,tw,t,t]
    V = torch.einsum("ij,ncabjk,lk->ncabil", BT, tiles, BT)   # B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.to(dt), G)      # G g G^T   [K,C,t,t]
    M = torch.einsum("ncabil,kcil->nkabil", V, U)             # sum over C per xi
    Y = torch.einsum("ij,nkabjl,ml->nkabim", AT, M, AT)       # A^T M A   [N,K,th,tw,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], th * m, tw * m)

for C, hw in ((64, 64), (256, 32), (512, 16)):
    x = torch.randn(2, C, hw, hw)
    x = torch.nn.functional.leaky_relu(x, 0.01)               # activations are mostly positive
    w = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    d32 = torch.nn.functional.conv2d(x, w, padding=1).double()
    sc = ref.abs().max()
    line = f"C={C:3d} {hw}x{hw}: direct fp32 max {float((d32 - ref).abs().max() / sc):.2e}"
    for kind in (2, 4):
        y = wino(x, w, kind).double()
        e = (y - ref).abs()
        line += f" | F({kind},3) max {float(e.max() / sc):.2e} rms {float(e.pow(2).mean().sqrt() / sc):.2e}"
    print(line)
