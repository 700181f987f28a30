// oracle/boost_shim - TEST INFRASTRUCTURE ONLY.  boost::serialization::make_array lives in serialization.hpp of this shim.
#pragma once
#include "serialization.hpp"
