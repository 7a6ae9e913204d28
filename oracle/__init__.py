"""
oracle -- CPU restatement (numpy / scipy.fft) of the pfb-imaging PCG / PSF-convolution /
wavelet hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is the checker, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``pfb_clean_amd/`` imports it, and the product path
raises if the HIP library is missing rather than falling back to this code.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
against golden vectors produced by executing the reference's own source files
(/root/reference/pfb/...) in the build container under stub modules for the
third-party packages that are not installed (numba, ducc0, pywt, numexpr, dask,
pyscilog); generator: tests/golden/make_golden.py, fixtures: tests/golden/*.npz.
The FFT underneath the reference is ducc0 (un-pinned in the reference's setup.py, not
installed here); scipy.fft (pocketfft, ducc's ancestor) stands in for it in both the
golden generator and this oracle, so the DFT itself is pinned by its mathematical
definition (tests/test_oracle_golden.py::test_conv_is_direct_circular_convolution).

Each function cites the reference file:line it restates.
"""
