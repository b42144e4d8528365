import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowconductor_amd import ops, transforms, utils, options
from flowconductor_amd.nn import nets
dev = torch.device('cuda:0')
torch.manual_seed(0)
t = transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(32, even=True), lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2)).eval().to(dev)
for lg in (15, 16, 17, 18, 19, 20):
    x = torch.randn(1 << lg, 32, device=dev)
    with torch.no_grad():
        for _ in range(3): t(x)
        torch.cuda.synchronize()
        with ops.KernelTimer("fc_affine_coupling_resnet") as k:
            for _ in range(10): t(x)
        torch.cuda.synchronize()
        ms = sorted(k.durations_ms())
        with options.override(fused_final_layer=False), ops.KernelTimer("fc_resnet_hidden") as kh, ops.KernelTimer("fc_affine") as ka:
            for _ in range(10): t(x)
        torch.cuda.synchronize()
    print("N=2^%d one kernel %.1f us | hidden %.1f us + affine %.1f us (+ GEMM)" % (lg, ms[len(ms)//2]*1e3, sorted(kh.durations_ms())[5]*1e3, sorted(ka.durations_ms())[5]*1e3))
