// oracle/boost_shim — TEST INFRASTRUCTURE ONLY.  Boost is not installed here; DBoW2's BowVector.h / FeatureVector.h only name
// boost::serialization::access (friend) and base_object<> inside never-instantiated serialize() templates.
#pragma once
namespace boost { namespace serialization {
class access;
template <class Base, class Derived> Base& base_object(Derived& d) { return static_cast<Base&>(d); }
} }
