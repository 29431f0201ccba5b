#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of one csrc/*.hip file (device-only compile, metadata of the gfx950 code object):
    python tools/kernel_regs.py k_gather_dot.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "dau-convnet_amd", "csrc", sys.argv[1])
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.dirname(src), "-mllvm", "-pragma-unroll-threshold=1000000", "--cuda-device-only", "-S",
                           src, "-o", os.path.join(d, "k.s")] + sys.argv[2:], stderr=subprocess.DEVNULL)
    txt = open(os.path.join(d, "k.s")).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void dau::", "")
    print("%-70s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (name[:70], get("vgpr_count"), blk.split()[0], get("sgpr_count"),
          get("vgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
