"""SparseMotionGenerator / SparseMotionDecoder: the object-graph GNN that predicts per-object affine thetas
(reference: src/modules/motion_estimator/sparse_motion_estimator.py:12-141).  Tiny FLOPs -> PyTorch device ops
(Linear/BatchNorm1d/GATv2 message passing), see SURVEY.md §2.4 "support".  The clicked node is drawn on the host with
NumPy like the reference unless the caller injects `click_index` (parity / graph-capture friendly)."""
import numpy as np
import torch
from torch import nn

from ...thirdparty import GATv2Conv


class SparseMotionGenerator(nn.Module):
    def __init__(self, num_features_x=7, num_features_y=2, z_dim=64, h_dim=64, num_head=4, input_scene_features=256,
                 h_scene_features=256, num_predicted_frames=5, num_input_frames=1):
        super().__init__()
        self.input_scene_features = input_scene_features
        self.h_scene_features = h_scene_features
        self.num_predicted_frames = num_predicted_frames
        self.num_input_frames = num_input_frames
        self.decoder = SparseMotionDecoder(h_dim, h_dim, z_dim, h_dim, num_predicted_frames, num_head)
        half = int(h_dim / 2)
        self.x_encoder = nn.Sequential(nn.Linear(num_features_x, half), nn.LeakyReLU(0.2), nn.Linear(half, h_dim))
        self.y_encoder = nn.Sequential(nn.Linear(num_features_y, half), nn.LeakyReLU(0.2), nn.Linear(half, h_dim))
        mid = int(input_scene_features / 2)
        self.encode_scene_features = nn.Sequential(
            nn.Linear((h_dim + input_scene_features) * self.num_input_frames, mid), nn.BatchNorm1d(mid),
            nn.LeakyReLU(0.2), nn.Linear(mid, h_dim * 2), nn.BatchNorm1d(h_dim * 2), nn.LeakyReLU(0.2),
            nn.Linear(h_dim * 2, h_dim))

    @staticmethod
    def draw_click_index(num_real_nodes, device):
        """One uniformly drawn real node per sample (sparse_motion_estimator.py:43-51); host-side NumPy RNG."""
        counts = [int(num_real_nodes)] if isinstance(num_real_nodes, int) else [int(n) for n in num_real_nodes.tolist()]
        picks, base = [], 0
        for n in counts:
            picks.append(int(np.random.randint(0, n)) + base)
            base += n
        return torch.tensor(picks, dtype=torch.long, device=device)

    def _encode(self, data, scene_features):
        x_map = self.x_encoder(data.x)
        # theta_map = y_encoder(targets_theta) of the reference (:60) feeds only the latent branch of the decoder, whose result
        # nothing reads (see SparseMotionDecoder.forward): not computed.  y_encoder / linear_z end up WITHOUT gradients in the
        # reference as well (the fixtures' "nograd" lists), so the observable state is the same.
        h = torch.cat(torch.unbind(torch.cat([x_map, scene_features], dim=2), 1), 1)
        return self.encode_scene_features(h), None

    def forward(self, data, scene_features, latent, click_index=None):
        if click_index is None:
            click_index = self.draw_click_index(data.num_real_nodes, data.x.device)
        u = torch.zeros(data.num_nodes, device=data.x.device)
        u.index_fill_(0, click_index, 1.0)               # u[click_index] = 1 without the host round trip of index_put_
        u = u.unsqueeze(1)
        h, theta_map = self._encode(data, scene_features)
        return self.decoder(h, data.x[:, :2], theta_map, data.edge_index, u, latent, data.targets_theta)

    def inference(self, data, z, index_user_guidance, scene_features):
        return self.forward(data, scene_features, z.to(data.x.device), click_index=index_user_guidance)


class SparseMotionDecoder(nn.Module):
    def __init__(self, num_features_x, num_features_y=2, z_dim=2, h_dim=64, num_predicted_frames=5, num_head=4):
        super().__init__()
        self.num_predicted_frames = num_predicted_frames
        self.z_dim = z_dim
        self.h_dim = h_dim
        self.act = nn.LeakyReLU(0.2)
        self.num_features_x = num_features_x
        self.num_features_y = num_features_y
        self.linear_z = nn.Sequential(nn.Linear(z_dim, h_dim * 2), nn.LeakyReLU(0.2), nn.Linear(h_dim * 2, h_dim))
        convs, locs = [], []
        for _ in range(num_predicted_frames):
            convs.append(GATv2Conv(num_features_x, num_features_x, add_self_loops=False, heads=num_head, concat=False))
            loc = nn.Sequential(nn.Linear(num_features_x, h_dim), nn.LeakyReLU(0.2), nn.Linear(h_dim, 3 * 2))
            loc[2].weight.data.zero_()                      # identity affine at init (:120-121)
            loc[2].bias.data.copy_(torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float))
            locs.append(loc)
        self.conv_time_steps = nn.ModuleList(convs)
        self.loc_time_steps = nn.ModuleList(locs)

    def forward(self, x_n, x_start_pos, y_n, edge_index, u_n, z, targets_theta):
        # (:127-128) the reference writes the latent branch  y_n[:, t] = linear_z(z[:, t]) * (1 - u_n) + y_n[:, t] * u_n  into y_n,
        # which nothing reads afterwards: ~60 launches per step for a discarded result (round 5: skipped; linear_z / y_encoder have
        # no gradient either way -- in the reference because nothing downstream of them reaches a loss).
        out, x = {}, x_n
        keep = 1 - u_n                                            # the two blend factors once, not per frame
        guided = targets_theta * u_n.unsqueeze(1)                 # [N, T, 6], carries no gradient
        for t in range(self.num_predicted_frames):
            x = self.conv_time_steps[t](x, edge_index)
            out[f"theta_{t}"] = self.loc_time_steps[t](x) * keep + guided[:, t, ...]
        return out
