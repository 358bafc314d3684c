"""PixelLink++ (drop-in package name for ssd_liverdet/pixel_link): model, criterion and link decoding on the HIP path."""
