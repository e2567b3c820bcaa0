"""sgmse_amd: MI355X-native reverse-SDE speech-enhancement sampler (hot path of sp-uhh/sgmse).

All arithmetic runs in hand-written HIP kernels (libsgmse_hip.so, gfx950) behind the C ABI of include/sgmse_hip.h;
this package is the host-side mirror of the reference's Python API for that path (ScoreModel.enhance / get_pc_sampler,
BackboneRegistry, SDERegistry, Predictor/CorrectorRegistry, SpecsDataModule front-end, pad_spec)."""
__version__ = "0.1.0"
