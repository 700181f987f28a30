// real_world.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  Force-included for the second matcher world
// (oracle/Makefile, _ref/libmw_ref_real.so and _ref/libmw_facade_real.so): ORBmatcher.cc / the facade run over the reference's OWN Frame
// (include/Frame.h + src/Frame.cc) and MapPoint (include/MapPoint.h + src/MapPoint.cc); only KeyFrame, Map, the camera model and the
// Eigen / Sophus algebra remain stand-ins.  The test driver has to set private state (poses, descriptors, distance limits, bad flags),
// so after every standard / third-party header has been seen normally, `private` and `protected` are opened for the reference's headers.
// Access specifiers do not change the layout g++ gives these classes, and every translation unit of the library is built the same way.
#ifndef ORBX_REAL_WORLD_H
#define ORBX_REAL_WORLD_H
#include <algorithm>
#include <fstream>
#include <iomanip>
#include <limits>
#include <numeric>
#include <sstream>
#include <string>
#include "mappoint_world.h"
#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"
#define private public
#define protected public
#endif
