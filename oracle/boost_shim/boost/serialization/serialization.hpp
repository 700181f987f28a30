// oracle/boost_shim — TEST INFRASTRUCTURE ONLY.  Boost is not installed here; DBoW2's BowVector.h / FeatureVector.h only name
// boost::serialization::access (friend) and base_object<> inside never-instantiated serialize() templates.
#pragma once
namespace boost { namespace serialization {
class access;
template <class Base, class Derived> Base& base_object(Derived& d) { return static_cast<Base&>(d); }
template <class T> int make_array(T*, unsigned long) { return 0; }          // include/KeyFrame.h:146, include/MapPoint.h:86 (inside serialize() templates)
} }
