"""configs[3] timing (tuning aid): python tools/probes/twostage_time.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
orig = bench.extra_configs


class A:
    pass


# run only the configs[3] part: monkeypatch by slicing is brittle, so time the whole extras and print configs[3]
out = orig(dev, A())
print(json.dumps(out["configs[3]"], indent=1))
