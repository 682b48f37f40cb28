#!/usr/bin/env python3
"""Identity of the device library as built from this tree: sha256 over the sources of libxmipp_hip.so (xmipp3_amd/csrc/*.hip, *.h, build.sh
and include/xmipp_hip.h), first 16 hex digits.  The collection scripts stamp their JSON with it ("library_source_sha16") and bench.py marks
side data collected from other sources "stale" (git is not available on the GPU box, the sources are).
    python3 tools/libhash.py            prints the hash
    python3 tools/libhash.py file.json  adds / replaces "library_source_sha16" and "xh_version" in that JSON document"""
import glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "xmipp3_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "xmipp3_amd", "csrc", "*.h")) +
                   [os.path.join(root, "xmipp3_amd", "csrc", "build.sh"), os.path.join(root, "include", "xmipp_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    s = source_hash()
    if len(sys.argv) > 1:
        d = json.load(open(sys.argv[1]))
        d["library_source_sha16"] = s
        d["xh_version"] = "xmipp3_amd 0.1 (gfx950)"
        json.dump(d, open(sys.argv[1], "w"), indent=1)
    print(s)
