from .ops import resample, get_grid, get_occlusion_map, mesh_grid  # noqa: F401
from .utils import (resize_flow, resize_video, isnan, set_random_seed, get_rank, is_master, get_world_size,  # noqa: F401
                    dist_all_reduce_tensor, dist_all_gather_tensor, init_cudnn)
