// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN, see opencv2/core/core.hpp in this directory.
#pragma once
#include "opencv2/core/core.hpp"
