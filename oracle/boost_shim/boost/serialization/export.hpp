// empty: names only referenced inside never-instantiated serialize() templates
#include "serialization.hpp"
