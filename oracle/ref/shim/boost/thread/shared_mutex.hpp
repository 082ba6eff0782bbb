// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN, see boost/thread.hpp in this directory.
#pragma once
#include "boost/thread.hpp"
