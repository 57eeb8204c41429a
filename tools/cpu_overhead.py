"""How much of a step is host time: wall time per step vs time the host spends issuing it (no sync inside the loop)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FULL = "--full" in sys.argv          # BASELINE configs[3]: both discriminators + the four Adam steps
if "--bf16" in sys.argv:
    from c2m_amd import ops
    ops.set_conv_precision("bf16")
cfg = normalize_config(default_config(height=128, width=256, num_input_frames=2, use_image_discriminator=FULL,
                                      use_video_discriminator=FULL))
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to("cuda:0").train()
step = TrainStep(model, run_optimizers=FULL, distributed=False)
batch = batch_to(make_batch(B, 128, 256, 2, seed=0), "cuda:0")
batch["rng"] = {k: v.to("cuda:0") for k, v in make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0).items()}
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
issue = []
for _ in range(10):
    a = time.perf_counter()
    step(batch)
    issue.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B}: wall {t_all / 10 * 1e3:.1f} ms/step; host issue time {t_issue / 10 * 1e3:.1f} ms/step "
      f"(per step: {[round(x * 1e3) for x in issue]})")

if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step(batch)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
