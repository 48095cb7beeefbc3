"""Static resources of every kernel of one source (LDS bytes per workgroup, VGPRs, AGPRs, occupancy, scratch) from
hipcc's -Rpass-analysis=kernel-resource-usage.   usage: python tools/resource_usage.py repo_amd/csrc/conv.hip [substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]


def field(b, f):
    m = re.search(f + r": (\d+)", b)
    return int(m.group(1)) if m else -1


print("#   LDS B   VGPR  AGPR  waves/SIMD  scratch  kernel")
for b in blocks:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"^void repo::", "", dn)
    if sub not in dn:
        continue
    print("%9d  %5d %5d  %10d  %7d  %s" % (field(b, r"LDS Size \[bytes/block\]"), field(b, "VGPRs"), field(b, "AGPRs"),
                                            field(b, r"Occupancy \[waves/SIMD\]"), field(b, r"ScratchSize \[bytes/lane\]"),
                                            dn[:130]))
