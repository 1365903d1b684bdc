#!/bin/bash
# Print VGPR / AGPR / SGPR / LDS / scratch use of every gfx950 kernel in a HIP object or shared library.
# usage: tools/kernel_regs.sh unet-implementations_amd/csrc/build/conv_patch.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $tmp/fat.bin
tgt=$($LLVM/clang-offload-bundler --list --type=o --input=$tmp/fat.bin | grep gfx950 | head -1)
$LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=$tgt --output=$tmp/dev.co
$LLVM/llvm-readelf --notes $tmp/dev.co | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    agpr = blk.split()[0]
    name = re.sub(r"\(anonymous namespace\)::|unet_conv::", "", g("name"))
    print("%-90s vgpr %s agpr %s sgpr %s lds %s scratch %s" % (name[:90], g("vgpr_count"), agpr, g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
'
rm -rf $tmp
