"""CPU oracle for the registration-and-fusion hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the
reference algorithm (maragall/image-stitcher) used as the checker in
``tests/``, in ``__graft_entry__.smoke()`` and as the timed CPU baseline in
``bench.py``.  Nothing under ``image-stitcher_amd/`` (the product) imports
it; the product path fails loudly when the HIP library is missing.

Pinning: the functions in ``oracle.registration`` were checked against the
imported reference (NumPy backend, scikit-image absent) by
``tests/golden/make_golden.py``; the expectations live in ``tests/golden``.
``oracle.fusion`` follows ``image_stitcher/stitcher.py`` as text (that module
is not importable without dask/zarr/ome_zarr) and is pinned by the
known-answer values of the reference's own ``stitcher_test.py``.
``oracle.subpixel`` restates scikit-image's upsampled-DFT registration, a
third-party dependency the reference does not vendor or pin:
PARITY UNPINNED for that module.
"""
