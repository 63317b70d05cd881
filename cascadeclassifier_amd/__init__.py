"""MI355X-native cascade-classifier hot path (integral images + Haar/LBP window evaluation) behind the reference's
own plugin surface. The compute lives in cascadeclassifier_amd/csrc (HIP, gfx950) behind include/cascadeclassifier_amd.h."""
from ._lib import CascadeError, LIB_PATH  # noqa: F401
from .detector import CascadeClassifier, group_rectangles, scale_plan  # noqa: F401
from .evaluator import CvFeatureEvaluator, CvFeatureParams, NegativeMiner  # noqa: F401
