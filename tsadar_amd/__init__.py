"""tsadar_amd -- MI355X-native Thomson-scattering form-factor engine (drop-in for the hot path of
ergodicio/tsadar: ThomsonScatteringDiagnostic.__call__ and LossFunction.vg_loss)."""
from .params import ThomsonParams  # noqa: F401
from .calibration import get_scattering_angles, sa_lookup  # noqa: F401

__all__ = ["ThomsonParams", "get_scattering_angles", "sa_lookup"]
