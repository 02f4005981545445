#!/usr/bin/env python3
"""One leg of bench.py's other_configs on its own:  python tools/lab/one_config.py cfg3 [cfg4 ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for line in bench.other_configs(torch, dev, only=set(sys.argv[1:] or ["cfg3"])):
    print(json.dumps(line), flush=True)
