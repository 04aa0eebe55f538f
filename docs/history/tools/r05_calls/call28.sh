#!/bin/bash
# round 5, call 28: how the HSA agent of the HIP device was found (PCI address vs "the only GPU"), PCI ids on both sides
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, subprocess
import strided_jl_amd as S
import numpy as np
p = torch.cuda.get_device_properties(0)
print("torch device 0:", p.name, getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
print(subprocess.run("rocminfo | grep -i -E 'BDFID|Domain|Device Type|Marketing' | head -20", shell=True, capture_output=True, text=True).stdout)
t = torch.zeros(32 ** 4, dtype=torch.float64, device="cuda")
from bench import colmajor_view
A = colmajor_view(S, t, (32,) * 4); B = colmajor_view(S, torch.zeros_like(t), (32,) * 4)
q = S.Sequence().add(S.make_plan(lambda x: x, None, None, A.size, (B, A.permutedims((3, 2, 1, 0)))))
print(q.info())
PY
