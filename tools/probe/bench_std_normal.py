import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops
x = torch.randn(1 << 20, 64, device='cuda')
lad = torch.randn(1 << 20, device='cuda')
for _ in range(3): ops.standard_normal_log_prob(x, 0.5 * 64 * 1.8378770664093453, lad)
with ops.KernelTimer("fc_standard_normal_log_prob") as t:
    for _ in range(20): out = ops.standard_normal_log_prob(x, 0.5 * 64 * 1.8378770664093453, lad)
torch.cuda.synchronize()
ms = sorted(t.durations_ms()); print("std normal median %.4f ms -> %.0f GB/s" % (ms[10], (x.numel()*4 + 8*(1<<20)) / ms[10] / 1e6))
ref = -0.5 * (x.double() ** 2).sum(1) - 0.5 * 64 * 1.8378770664093453 + lad.double()
print("max err", float((out.double() - ref).abs().max()))
