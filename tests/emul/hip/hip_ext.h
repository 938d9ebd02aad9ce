// TEST-ONLY stand-in for <hip/hip_ext.h>; see hip_runtime.h in this directory.
#pragma once
#include "hip_runtime.h"
