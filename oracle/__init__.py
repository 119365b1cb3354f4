"""CPU oracle for the saliency-attribution hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (NumPy arithmetic; the
classifier stays an opaque torch callable) of the reference algorithms the HIP path
replaces.  It may be imported only by `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py`, and there only as the checker / the thing timed as a
CPU baseline -- never by anything under `image-classification-xai_amd/` (the product).

Pinning (how we know the oracle equals the reference):
  * IG / Left-IG / IDG / IDGI / slopes / alpha schedule, gkern / blur / auc, the pixel
    order, every perturbed image and the return tuples of the five ins/del metric
    classes are checked in `tests/test_oracle_golden.py` against `tests/golden/*.npz`,
    vectors produced by importing the reference itself (`tests/golden/make_golden.py`).
  * Grad-CAM (captum 0.7.0 `LayerGradCam`, a dependency that is absent from the reference
    tree and from this image) and the RISE mask generator (`skimage.transform.resize`,
    absent, version unpinned by the reference's requirements.txt) are restated from
    their published algorithms: PARITY UNPINNED for those two (see oracle/gradcam.py,
    oracle/rise.py).

The reference is Python, so the oracle is Python/NumPy; there is no C to compile.
"""
