// TEST INFRASTRUCTURE: puts the drop-in extractor facade in front of the reference's include/ORBextractor.h for the
// _ref/libref_frame_dropin.so build, so that the reference's own Frame.cc (#include "ORBextractor.h") is compiled against the facade, as
// INTEGRATION.md §2 tells a maintainer to do.
#include "../../../include/orb_slam3_amd/ORBextractor.h"
