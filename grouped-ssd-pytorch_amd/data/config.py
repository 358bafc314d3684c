"""Prior-box configurations of the path (values restated from ssd_liverdet/data/config.py:91-134)."""

# SSD300, the configuration build_ssd uses (config.py:114-134)
v2 = {
    'feature_maps': [38, 19, 10, 5, 3, 1],
    'min_dim': 300,
    'steps': [8, 16, 32, 64, 100, 300],
    'min_sizes': [30, 60, 111, 162, 213, 264],
    'max_sizes': [60, 111, 162, 213, 264, 315],
    'aspect_ratios': [[2], [2, 3], [2, 3], [2, 3], [2], [2]],
    'variance': [0.1, 0.2],
    'clip': True,
    'name': 'v2',
}

# SSD512 geometry (config.py:91-110); priors only -- the 512 model variant is dead code upstream
v2_512 = {
    'feature_maps': [64, 32, 16, 8, 4, 2, 1],
    'min_dim': 512,
    'steps': [8, 16, 32, 64, 128, 256, 512],
    'min_sizes': [20, 51, 133, 215, 296, 378, 460],
    'max_sizes': [51, 133, 215, 296, 378, 460, 542],
    'aspect_ratios': [[2], [2, 3], [2, 3], [2, 3], [2, 3], [2], [2]],
    'variance': [0.1, 0.2],
    'clip': True,
    'name': 'v2_512',
}
