#!/usr/bin/env python3
"""The inference leg of bench.py on its own (ensemble_outputs on 60-s files, trainv2.py:158-192 = evaluator.py:16-50): prints its record; run under
rocprofv3 --kernel-trace --stats to see what a file's 1.7 ms are made of."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
print(json.dumps(bench.inference_leg(dev, 0, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 271)))
