"""Network builders: GSSD / GSSD++ (ssd_multiphase_custom_group) and the vanilla SSD300 (ssd)."""
