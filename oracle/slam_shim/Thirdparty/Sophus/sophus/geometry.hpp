// TEST INFRASTRUCTURE: resolves `#include "Thirdparty/Sophus/sophus/geometry.hpp"` of the reference's include/Frame.h:28 (this directory
// precedes the reference root on the include path) to the stand-in types of slam_types.h.
#include "../../../slam_types.h"
