// TEST INFRASTRUCTURE: resolves `#include "sophus/sim3.hpp"` of the reference's include/ORBmatcher.h:26 to the stand-in types of
// oracle/slam_shim/slam_types.h (the real Sophus needs Eigen, which is not installed here).
#include "../slam_types.h"
