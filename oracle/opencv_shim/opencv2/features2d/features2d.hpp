// oracle shim: forwards to the single minimal header (test infrastructure only)
#pragma once
#include "../opencv.hpp"
