import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from ilqgames_amd import abi, examples, hip
for cfg in ("mixed_dubins_car_scene", "three_unicycle_scene", "mixed_dubins_car_scene_open_loop"):
    spec = examples.CONFIGS[cfg]()
    x0 = examples.jittered_x0(spec, 256, seed=0)
    for kw in (dict(), dict(fixed_iters=6)):
        a = hip.Problem(spec, abi.F64).solve(x0, probe=False, **kw)
        b = hip.Problem(spec, abi.F64).solve(x0, probe=True, **kw)
        torch.cuda.synchronize()
        same = all(torch.equal(a[k], b[k]) for k in ("iters", "status", "converged", "xs", "us", "P", "alpha", "costs"))
        print(cfg, kw, "bit-identical:", same, "iters", int(a["iters"].sum()), int(b["iters"].sum()), "status", int(a["status"].sum()), int(b["status"].sum()))
        if not same:
            d = (a["iters"] != b["iters"]).nonzero().flatten().cpu().numpy()
            print("   differing instances", d[:10], a["iters"][d[:5]].cpu().numpy(), b["iters"][d[:5]].cpu().numpy(), a["status"][d[:5]].cpu().numpy(), b["status"][d[:5]].cpu().numpy())
