// Stand-in for <hip/hip_runtime.h> when the engine's arithmetic headers (fe.cuh, sc.cuh, ge.cuh, keccak.cuh) are
// compiled for the HOST by tests/hostsim/arith_host.cpp: the qualifiers vanish, nothing else is needed.
#pragma once
#include <stdint.h>
#include <string.h>
#define __device__
#define __constant__
#define __global__
#define __host__
#define __forceinline__ inline
#define __restrict__
