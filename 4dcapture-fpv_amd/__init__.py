"""fdcap_amd -- MI355X-native accelerator for the global body-scene optimisation hot path of
aptx4869lm/4DCapture-FPV (`global_optimization.py`, FittingOP.fitting(mode='global')).

Layout:
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/fdcap.h) -> libfdcap_hip.so
  capi.py      ctypes binding of the C-ABI (tensors -> raw device pointers + stream)
  ops.py       drop-in operators with the third-party call signatures the reference uses
               (chamferDist, body model, VPoser decode)
  fitting.py   FittingOP mirror (init / fitting / save_result) driving the fused HIP iteration
               (modes 'global', 'local', 'dct')
  smoother.py  optimization.py's per-frame smoother (FittingOP.fitting / fitting_smoothing) in one launch
  innerfit.py  per-frame inner fit with a 2D-keypoint reprojection term (outside the reference)
  io.py        body_gen -> smoothed_body pickle interface, camerapose.txt, scene readers
  synth.py     seeded synthetic stand-ins for the licensed assets
  dist.py      frame sharding + halo exchange over torch.distributed (RCCL on ROCm)
"""
__all__ = ["synth"]
