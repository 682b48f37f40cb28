"""Register / LDS use of every kernel of a .hip file (clang's kernel-resource-usage remarks): python tools/kernel_resources.py xh_pm.hip [extra flags]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xmipp3_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(here, src), "-o", "/tmp/_res.o"] + sys.argv[2:]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].split(" ")[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-72s VGPR %4d AGPR %4d SGPR %4d LDS %6d scratch %5d occupancy %d" % (name[-72:], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"LDS Size \[bytes/block\]"),
                                                                     g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
