#!/usr/bin/env python3
"""640 uint8 images through the HIP Inception extractor, a few times: run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd.inception import InceptionV3  # noqa: E402

m = InceptionV3().to("cuda:0")
m.weights_loaded = True
u8 = torch.randint(0, 256, (640, 64, 64, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(5):
    m.features(u8)
torch.cuda.synchronize()
