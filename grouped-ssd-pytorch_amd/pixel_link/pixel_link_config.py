"""Module-level constants of ssd_liverdet/pixel_link/pixel_link_config.py the path reads (same names, same values)."""
version = "4s"              # :1   ("2s" -- a fifth 150x150 output stage -- is not built in the HIP path)
dilation = True             # :4
pixel_weight = 2            # :21
link_weight = 1             # :20
neg_pos_ratio = 3           # :22
min_area = 3                # :23
min_height = 1              # :24
pixel_conf_threshold = 0.2  # :27
link_conf_threshold = 0.8   # :28
vgg_groups = 4              # :31
feature_scale = 1           # :32
image_height = 300
image_width = 300
