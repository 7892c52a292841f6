#!/usr/bin/env python3
"""Reference point: which library kernels torch's fp32 GEMM dispatches to on the model's shapes (run under rocprofv3 --kernel-trace)."""
import torch
for (m, n, k) in [(7840, 2048, 512), (7840, 512, 2048), (7840, 1536, 512), (1960, 3072, 768), (1960, 768, 3072), (7840, 512, 512),
                  (1568, 1536, 384), (1568, 384, 1536), (125440, 512, 128), (31360, 1024, 256)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(k, n, device="cuda"); b = torch.randn(n, device="cuda")
    for _ in range(3):
        torch.addmm(b, x, w)
    torch.cuda.synchronize()
