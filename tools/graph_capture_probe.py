"""Probe: capture one G-only training step (zero_grad + forward + backward) of the tiny network into a HIP graph, replay it and
compare losses / gradients with the eager step.  usage: python tools/graph_capture_probe.py [fwd|step]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else "step"
if len(sys.argv) > 2 and sys.argv[2] == "rocblas":
    # hypothesis: the hipBLASLt "UserArgs" GEMM kernels (nn.Linear of the GNN / fc layers) take their argument block from a
    # host-staged copy that a graph replay re-reads after it has gone stale
    torch.backends.cuda.preferred_blas_library("cublas")
cfg = normalize_config(default_config(num_input_frames=2, block_expansion=4, max_expansion=32, h_dim=32, z_dim=16, out_channel=16,
                                      ndf=4, use_image_discriminator=False, use_video_discriminator=False))
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").cuda().train()
batch = batch_to(make_batch(1, 128, 256, 2, seed=0), "cuda:0")
rng = make_step_rng(batch, z_dim=16, latent_dim=32, seed=0)
batch["rng"] = {k: v.cuda() for k, v in rng.items()}
step = TrainStep(model, run_optimizers=False, distributed=False)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        out, lg, _ = step(batch)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
ref_total = float(lg["total_gen"])
ref_grad = model.generator.first.conv.weight.grad.clone()
print("eager total", ref_total, flush=True)
g = torch.cuda.CUDAGraph()
for p in model.parameters():
    p.grad = None
with torch.cuda.graph(g, stream=side):
    if mode == "fwd":
        with torch.no_grad():
            out_g, lg_g, _, _ = model(batch)
        total_g = sum(v for v in lg_g.values())
    else:
        out_g, lg_g, _ = step(batch)
        total_g = lg_g["total_gen"]
print("captured", flush=True)
torch.cuda.synchronize()
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(total_g), flush=True)
if mode != "fwd":
    gg = model.generator.first.conv.weight.grad
    print("grad max diff", float((gg - ref_grad).abs().max()), "total diff", abs(float(total_g) - ref_total))
print("GRAPH-PROBE-OK")
