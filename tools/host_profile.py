"""cProfile of the host side of one training step (B=1 so that the GPU never throttles the host)."""
import copy, cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c2m_amd.config import default_config, normalize_config
from c2m_amd.modules.model import GeneratorFullModel
from c2m_amd.synthetic import make_batch, make_step_rng, batch_to
from c2m_amd.train import TrainStep

cfg = normalize_config(default_config(height=128, width=256, num_input_frames=2, use_image_discriminator=False,
                                      use_video_discriminator=False))
torch.manual_seed(0)
model = GeneratorFullModel(train_params=copy.deepcopy(cfg)["train_params"], model_params=copy.deepcopy(cfg)["model_params"],
                           dataset="cityscapes").to("cuda:0").train()
step = TrainStep(model, run_optimizers=False, distributed=False)
batch = batch_to(make_batch(1, 128, 256, 2, seed=0), "cuda:0")
batch["rng"] = {k: v.to("cuda:0") for k, v in make_step_rng(batch, z_dim=1024, latent_dim=1024, seed=0).items()}
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
