"""diffphore_amd - MI355X-native (gfx950) implementation of DiffPhore's reverse-diffusion hot path.

Public surface mirrors the reference for this path only:
    TensorProductScoreModel.forward(data) -> (tr_pred, rot_pred, tor_pred)     src/models/score_model_phore.py:294-310
    sampling_phore(...), randomize_position(...)                                src/utils/sampling.py:16-63,174-280
    t_to_sigma, get_t_schedule, set_time_phore                                  src/utils/diffusion_utils.py
All arithmetic runs in the HIP library `diffphore_amd/csrc/libdiffphore_hip.so` behind the C ABI in
`include/diffphore_amd.h`; there is no CPU or PyTorch fallback.
"""
__version__ = "0.1.0"
