import sys, os, cProfile, pstats, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from flowconductor_amd import transforms, utils, flows, distributions
from flowconductor_amd.nn import nets
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = [transforms.AffineCouplingTransform(utils.create_alternating_binary_mask(32, even=(i % 2 == 0)), lambda a, b: nets.ResidualNet(a, b, hidden_features=64, num_blocks=2)) for i in range(8)]
flow = flows.Flow(transforms.CompositeTransform(layers), distributions.StandardNormal([32])).eval().to(dev)
x = torch.randn(4096, 32, device=dev)
with torch.no_grad():
    for _ in range(20): flow.log_prob(x)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200): flow.log_prob(x)
    torch.cuda.synchronize()
    print("per log_prob call: %.1f us (8 layers)" % ((time.perf_counter() - t0) / 200 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): flow.log_prob(x)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
