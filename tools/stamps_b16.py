#!/usr/bin/env python3
"""Phase stamps (s_memrealtime, 10 ns ticks) of conv_patch_b16_kernel for one fused forward layer.
Needs the -DB16_STAMPS build:  make BUILD=build_st OUT=../libunet_st.so EXTRA=-DB16_STAMPS
Usage: UNET_HIP_LIB=$PWD/unet-implementations_amd/libunet_st.so python tools/stamps_b16.py Cin Cout H"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
Cin, Cout, H = (int(v) for v in sys.argv[1:4])
N, BF = 8, torch.bfloat16
x = torch.randn(N, H, H, Cin, device="cuda").to(BF)
al = torch.rand(N, Cin, device="cuda") + 0.5
be = torch.randn(N, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
b = torch.zeros(Cout, device="cuda")
g1 = torch.ones(Cout, device="cuda"); b1 = torch.zeros(Cout, device="cuda")
table = ops.PackTable([w], True, None); table.run()
src = ops.Act(x, al, be)
for _ in range(3):
    ops.conv_in_fwd(src, None, 0.01, table.wf[0], b, 3, 1, g1, b1, 1e-5, None, b16=True, w3=table.wf3[0])
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["UNET_HIP_LIB"])
buf = (ctypes.c_ulonglong * (256 * 16))()
assert lib.unet_debug_stamps(buf) == 0
import statistics
rows = []
for blk in range(256):
    t = [buf[blk * 16 + i] for i in range(16)]
    n = max(i for i in range(16) if t[i]) + 1 if any(t) else 0
    if n >= 4:
        rows.append([ (t[i] - t[0]) * 10 for i in range(n)])
print("stamps per block (ns since the block's start): start, loads issued, first stage in LDS, after each K step, output stored, end")
n = min(len(r) for r in rows)
print("median over", len(rows), "blocks:", [int(statistics.median(r[i] for r in rows)) for i in range(n)])
for r in rows[:6]:
    print(r)
t0 = min(buf[blk * 16] for blk in range(256) if buf[blk*16]); t1 = max(max(buf[blk*16+i] for i in range(16)) for blk in range(256))
print("first 256 blocks: span", (t1 - t0) * 10, "ns")
