"""MI355X-native batched decentralized EKF + MHE estimator (hot path of
well-robotics/Decentralized_EKF_MHE).  The compute path is hand-written HIP for gfx950
behind the C ABI in include/dekf.h; this package is the thin host-side mirror."""
from .params import DekfParams, go1_params, cassie_params, pogox_params  # noqa: F401
