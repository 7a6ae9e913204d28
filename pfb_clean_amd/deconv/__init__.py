"""CLEAN minor cycles that call the PSF convolution (pfb/deconv)."""
