// pbh_config.hpp -- precision selection.  The kernel and host sources are compiled twice:
//   default      : float32 arithmetic on complex64 data, 2^14-point tiles, 32 points per thread
//   -DPBH_F64    : float64 arithmetic on complex128 data, 2^13-point tiles, 16 points per thread
// (same LDS bytes and the same VGPR footprint per thread in both builds).  Everything lives in
// namespace PBH_NS and exports C symbols with the PBH_FN prefix; pbhip_api.cpp dispatches the public
// pbh_* ABI on the plan's dtype.  The reference accepts both dtypes (pulsarbat/core.py:742) and
// keeps dtype in = dtype out (tests/test_fft.py:53-54).
#pragma once
#include <hip/hip_runtime.h>

#define PBH_CAT2(a, b) a##b
#define PBH_CAT(a, b) PBH_CAT2(a, b)

#ifdef PBH_F64
#define PBH_NS pbh64
#define PBH_PREFIX pbh64_
#define PBH_REAL double
#define PBH_REAL2 double2
#define PBH_MAKE2 make_double2
#define RC(x) x
#define PBH_TILE_LOG2 13
#define PBH_R 16
#define PBH_LOG2R 4
#else
#define PBH_NS pbh32
#define PBH_PREFIX pbh32_
#define PBH_REAL float
#define PBH_REAL2 float2
#define PBH_MAKE2 make_float2
#define RC(x) x##f
#define PBH_TILE_LOG2 14
#define PBH_R 32
#define PBH_LOG2R 5
#endif
#define PBH_FN(name) PBH_CAT(PBH_PREFIX, name)
