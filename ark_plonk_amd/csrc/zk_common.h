// Common macros for code shared between host (final MSM combine, domain constants) and gfx950 device code.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_D __device__ __forceinline__
#else
#define ZK_HD inline
#define ZK_D inline
#endif
