"""MI355X-native ORB front-end for ORB-SLAM3 (extract + match): hand-written HIP kernels for gfx950 behind
the reference's ORBextractor / ORBmatcher interfaces.  See DESIGN.md and include/orbx.h."""
from ._lib import KP_DTYPE, OrbxError, load_hip  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import (ORBmatcher, ComputeStereoMatches, StereoFishEyeKnn, GetFeaturesInArea, AreaSearchBatch,  # noqa: F401
                      ComputeDistinctiveDescriptors)
from .vocabulary import ORBVocabulary  # noqa: F401
from . import views, sophus  # noqa: F401
from .sophus import SE3f, Sim3f  # noqa: F401
