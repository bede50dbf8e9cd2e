"""Host-side mirrors of the reference's ``src/model`` hot-path modules."""
from .dual_stream_attention import (DualStreamCrossAttention, MOUTH_INDICES, EXPRESSION_INDICES,
                                    ARKIT_BLENDSHAPES, MOUTH_BLENDSHAPES)
from .simplified_dual_stream_model import SimplifiedDualStreamModel
from .sequential_dual_stream_model import SequentialDualStreamModel
from .simplified_model import SimplifiedKoeMorphModel
from .gaussian_face import KoeMorphModel, create_koemorph_model

__all__ = ["DualStreamCrossAttention", "SimplifiedDualStreamModel", "SequentialDualStreamModel", "SimplifiedKoeMorphModel",
           "KoeMorphModel", "create_koemorph_model",
           "MOUTH_INDICES", "EXPRESSION_INDICES", "ARKIT_BLENDSHAPES", "MOUTH_BLENDSHAPES"]
