#!/usr/bin/env python3
"""Print the measured parity numbers (HIP engine vs golden fixtures) -- GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import numpy as np  # noqa: E402
from gpu_common import run_fixture, wh_error  # noqa: E402
from conftest import solver_fixture_names  # noqa: E402


def main():
    import importlib
    only = sys.argv[1:] or None
    for name in solver_fixture_names():
        method = {"mur": "mur", "aoadmm": "ao_admm", "admm": "admm", "anls": "anls"}[name.split("_")[0]]
        if only and not any(name.startswith(o) for o in only):
            continue
        solver = getattr(importlib.import_module("nmf_amd." + method), method, None)
        try:
            z, meta, v, res = run_fixture(name, solver)
        except Exception as e:  # noqa: BLE001
            print(f"{name:30s} ERROR {type(e).__name__}: {e}")
            continue
        err = wh_error(res.w, res.h, z["w"], z["h"], v)
        oh = np.asarray(res.obj_history)
        ref = z["obj_history"]
        nn = min(len(oh), len(ref))
        orel = np.max(np.abs(oh[:nn] - ref[:nn]) / np.maximum(np.abs(ref[:nn]), 1e-300))
        print(f"{name:30s} i={res.i:4d} (ref {int(z['i']):4d})  WH err {err:.3e}  obj max rel {orel:.3e}")


if __name__ == "__main__":
    main()
