#!/usr/bin/env python3
"""(needs a library built with XH_DEBUG_HOOKS=1 xmipp3_amd/csrc/build.sh: the XH_ES_ORDER hook is not in the product build)
Where xh_iterative_alignment and the oracle's chain part ways: per order (RS / SR) and per number of rounds, the
images whose pose differs, and whether the two orders' merits of such an image are within float rounding of each other (then the final
"better merit" choice is a coin toss between two valid poses).  Run on the GPU box: python3 tools/diag_iterative.py"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def child(order, draw, n, iters):
    import torch
    import xmipp3_amd as xa
    from oracle import pyoracle as o
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_estimators import _es_population
    D, sh, rot, ref, others = _es_population(o, draw, n, False)
    ctx = xa.Context(0)
    ms = min(20, D // 2 - 1)
    poses, merit = xa.iterative_alignment(ctx, torch.from_numpy(ref).cuda(), torch.from_numpy(others).cuda(), ms, iters)
    m = min(n, 40)
    ep, em = o.es_iterative_alignment(ref, others[:m], ms, iters, order=order if order != "both" else None)
    print(json.dumps({"D": D, "poses": poses[:m].tolist(), "merit": merit[:m].tolist(), "eposes": ep.tolist(), "emerit": em.tolist()}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    for draw in (2, 4, 6):
        for iters in (1, 3):
            res = {}
            for order in ("RS", "SR", "both"):
                env = dict(os.environ)
                if order != "both":
                    env["XH_ES_ORDER"] = order
                r = subprocess.run([sys.executable, __file__, order, str(draw), "100", str(iters)], env=env, capture_output=True, text=True)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print("failed", order, r.stderr[-300:]); continue
                res[order] = json.loads(line[-1])
            if len(res) < 3:
                continue
            D = res["both"]["D"]
            same = {k: np.array([np.allclose(np.array(v["poses"][i]), np.array(v["eposes"][i]), rtol=0, atol=1e-5) for i in range(len(v["poses"]))]) for k, v in res.items()}
            print(f"draw {draw} D {D} iters {iters}: equal poses RS {same['RS'].mean():.3f} SR {same['SR'].mean():.3f} chosen {same['both'].mean():.3f}")
            for i in np.nonzero(~same["both"])[0]:
                mrs, msr = res["RS"]["merit"][i], res["SR"]["merit"][i]
                ers, esr = res["RS"]["emerit"][i], res["SR"]["emerit"][i]
                print(f"   image {i}: device merits RS {mrs:.7f} SR {msr:.7f} (diff {mrs - msr:+.2e}); oracle RS {ers:.7f} SR {esr:.7f} (diff {ers - esr:+.2e}); "
                      f"per-order poses equal: RS {bool(same['RS'][i])} SR {bool(same['SR'][i])}")
