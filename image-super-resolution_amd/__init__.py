"""MI355X-native engine for the FreqFusionSR x4 inference hot path.

The directory name contains a hyphen (it mirrors the reference repository's name), so import it with
``importlib.import_module("image-super-resolution_amd")``.  Layout:

  csrc/            hand-written HIP kernels (gfx950) + the C ABI declared in include/ffsr.h
  hip.py           ctypes binding of libffsr_hip.so (fails loudly when the library is missing)
  ops.py           launch helpers on channels-last torch tensors (torch = device memory + streams only)
  nafnet.py drct.py grl.py mambair.py   the four frozen experts, host side (reference state_dict keys)
  fusion.py        the 7-phase frequency-guided fusion network
  engine.py        per-image pipeline of models/team29_FreqFusionSR/io.py (_load_all_models, _process_image)
  weights.py       reference-compatible random initialisation and checkpoint conventions
  shard.py         one-process-per-GPU image sharding + RCCL weight broadcast
"""
__all__ = ["hip", "ops", "weights", "nafnet", "drct", "grl", "mambair", "fusion", "engine", "shard"]
