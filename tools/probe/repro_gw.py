import copy, sys, os, torch
sys.path[:0] = [os.getcwd()]
from flowconductor_amd import transforms as T, utils
from flowconductor_amd.nn import nets
from oracle import torch_oracle as O

def run(d, hidden, k, blocks, maskk, n, scale, seed=0):
    torch.manual_seed(seed)
    mask = utils.create_alternating_binary_mask(d, even=True) if maskk == 0 else utils.create_mid_split_binary_mask(d)
    t = T.PiecewiseRationalQuadraticCouplingTransform(mask, lambda i, o: nets.ResidualNet(i, o, hidden_features=hidden, num_blocks=blocks), num_bins=k, tails="linear", tail_bound=3.0)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(scale)
    x = torch.randn(n, d, generator=g) * 1.3
    gy, gl = torch.randn(n, d, generator=g), torch.randn(n, generator=g)
    def grads(dtype, device):
        tt = copy.deepcopy(t).to(dtype).to(device).train()
        xi = x.to(dtype).to(device).clone().requires_grad_(True)
        y, lad = (O.transform_apply(tt, xi) if device == "cpu" else tt(xi))
        ((y * gy.to(dtype).to(device)).sum() + (lad * gl.to(dtype).to(device)).sum()).backward()
        return {nm: p.grad.detach().cpu().double() for nm, p in tt.named_parameters()}
    r = grads(torch.float64, "cpu"); f = grads(torch.float32, "cpu"); h = grads(torch.float32, "cuda")
    nm = "transform_net.final_layer.weight"
    P = 3 * k - 1
    e = (h[nm] - r[nm]).abs(); ef = (f[nm] - r[nm]).abs()
    dt = e.shape[0] // P
    per_dim = e.view(dt, P, -1).amax(dim=(1, 2))
    print("d %d hidden %d k %d blocks %d mask %d n %d scale %.1f: gW err %.3e (f32 oracle %.3e) scale %.2f; worst dims %s; per-param-row worst %s; bias err %.2e" % (
        d, hidden, k, blocks, maskk, n, scale, e.max(), ef.max(), r[nm].abs().max(), torch.topk(per_dim, 3).indices.tolist(),
        torch.topk(e.view(dt, P, -1).amax(dim=(0, 2)), 3).indices.tolist(), (h["transform_net.final_layer.bias"] - r["transform_net.final_layer.bias"]).abs().max()))

run(70, 32, 10, 2, 1, 1443, 1.8)
run(70, 32, 10, 2, 1, 1443, 1.0)
run(70, 32, 10, 2, 1, 256, 1.8)
run(70, 64, 10, 2, 1, 1443, 1.8)
run(64, 32, 10, 2, 1, 1443, 1.8)
run(74, 16, 8, 1, 0, 1098, 1.8)
run(64, 64, 8, 2, 0, 4096, 1.8)
run(64, 64, 8, 2, 0, 4096, 1.0)
