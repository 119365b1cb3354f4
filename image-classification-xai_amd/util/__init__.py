"""Drop-in mirror of the reference's `util` package for the accelerated hot path only
(attribution_methods.saliencyMethods, attribution_methods.CLIP.generate_emap [RISE part],
test_methods.*, model_utils).  Put `image-classification-xai_amd/` first on sys.path and
the reference harness imports resolve here; see INTEGRATION.md."""
