#!/usr/bin/env python3
"""Run chosen legs of bench.py's other_configs alone:  python tools/lab/bench_legs.py aoadmm_kl_on_cfg3_shape admm_kl_on_cfg3_shape"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NMF_AMD_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for c in bench.other_configs(torch, dev, only=set(sys.argv[1:])):
    c.pop("kernels", None) if os.environ.get("BRIEF") else None
    print(json.dumps(c)[:3000], flush=True)
