"""rectools_amd — MI355X-native engine behind RecTools' SASRec/BERT4Rec/HSTU fit()+recommend() hot path.

Hand-written gfx950 HIP kernels behind a C ABI (`include/rectools_hip.h`, `rectools_amd/csrc/`), driven from
Python/PyTorch-ROCm through the same plug-in interfaces the reference exposes (`Ranker`, transformer layers,
similarity module, lightning-module training step, model `fit`/`recommend`).  See DESIGN.md.
"""
__version__ = "0.1.0"
