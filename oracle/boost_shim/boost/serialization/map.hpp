#pragma once
#include "serialization.hpp"
