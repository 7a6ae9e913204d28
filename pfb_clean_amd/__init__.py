"""
pfb_clean_amd -- MI355X (gfx950) implementation of the pfb-imaging PCG / PSF-convolution /
wavelet hot path behind the reference's own Python entry points:

    pfb.operators.psf      -> pfb_clean_amd.operators.psf      psf_convolve_slice/cube
    pfb.operators.hessian  -> pfb_clean_amd.operators.hessian  _hessian_psf_slice, hessian_psf_cube
    pfb.operators.psi      -> pfb_clean_amd.operators.psi      Psi
    pfb.opt.pcg            -> pfb_clean_amd.opt.pcg            pcg, pcg_psf
    pfb.opt.power_method   -> pfb_clean_amd.opt.power_method   power_method
    pfb.opt.primal_dual    -> pfb_clean_amd.opt.primal_dual    primal_dual_optimised
    pfb.prox.prox_21m      -> pfb_clean_amd.prox.prox_21m      prox_21m_numba, dual_update_numba
    pfb.utils.misc         -> pfb_clean_amd.utils.misc         norm_diff, l1reweight_func

All arithmetic runs in hand-written HIP kernels (libpfb_hip.so, C-ABI in
include/pfb_hip.h).  There is no CPU fallback: importing the operator modules without
the built library, or calling them without a ROCm device, raises.
"""
__version__ = '0.1.0'
