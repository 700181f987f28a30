// TEST INFRASTRUCTURE: include/LoopClosing.h:31 includes <boost/algorithm/string.hpp> and uses nothing of it (compile check of LoopClosing.cc,
// oracle/slam_shim/loopclosing_world.h).  Boost is an absent external dependency.
#pragma once
