// oracle/boost_shim - TEST INFRASTRUCTURE ONLY.  include/MapPoint.h:86-87 names boost::serialization::make_array inside a never-instantiated
// serialize() template.
#pragma once
#include <cstddef>
namespace boost { namespace serialization {
template <class T> int make_array(T*, std::size_t) { return 0; }
} }
