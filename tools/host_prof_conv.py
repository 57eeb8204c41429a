import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from c2m_amd import ops
ops.set_conv_precision("bf16")
dev = "cuda:0"
x = torch.randn(8, 64, 32, 64, device=dev).bfloat16().requires_grad_(True)
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
b = torch.nn.Parameter(torch.zeros(64, device=dev))
def it():
    y = ops.conv(x, w, b, 1, 1, padding_mode="reflect", act="lrelu")
    y.float().sum().backward()
for _ in range(20): it()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): it()
host = time.perf_counter() - t
torch.cuda.synchronize()
print("host us per fwd+bwd:", host / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): it()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
