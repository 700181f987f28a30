// empty: names only referenced inside never-instantiated serialize() templates (boost::serialization::base_object is declared in serialization.hpp)
#include "serialization.hpp"
