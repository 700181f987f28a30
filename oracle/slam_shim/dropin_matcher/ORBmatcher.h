// TEST INFRASTRUCTURE: puts the drop-in facade in front of the reference's include/ORBmatcher.h for the libmw_facade_real.so build, so
// that the reference's own Frame.cc and MapPoint.cc (which #include "ORBmatcher.h") are compiled against the facade, as INTEGRATION.md §4
// tells a maintainer to do.
#include "../../../include/orb_slam3_amd/ORBmatcher.h"
