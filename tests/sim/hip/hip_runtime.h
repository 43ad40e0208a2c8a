// Stand-in for <hip/hip_runtime.h> when the kernel sources are compiled for the wave64 functional simulator (tests/sim/README.md).
#pragma once
#include "../sim_rt.h"
