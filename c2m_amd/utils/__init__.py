from .ops import resample, get_grid, grid_sample, get_occlusion_map, get_corresponding_map, mesh_grid  # noqa: F401
from .utils import resize_flow, resize_video, isnan, set_random_seed, get_rank, is_master  # noqa: F401
