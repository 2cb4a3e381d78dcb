// host_asan.cpp — the HOST code of librt355 that parses untrusted input or indexes by computed offsets, built with plain g++ under
// the address and undefined-behaviour sanitizers (no GPU code, no HIP runtime calls): the OpenEXR reader (rt_exr.cpp, compiled next
// to this file) and the host twins of the sphere grid, the direction-mask table and the candidate lists (rt_host_twins.h).
//   make -C rsraytracingv3_amd/csrc host-asan   ->  ../librt355_host_asan.so
//   tests/test_sanitized_host.py runs the CPU-tier tests of those parts against it in a child process (LD_PRELOAD=libasan).
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <cstring>
#include <vector>

#include "../../include/rt355.h"
#include "rt_device.h"
#include "rt_grid.h"

using namespace rt355;

extern "C" {
uint32_t rt_abi_version(void) { return RT355_ABI_VERSION; }
int rt_has_ab_shapes(void) { return 0; }
// (rt_set_skybox_exr, in rt_exr.cpp, hands the decoded image to the device side: there is none in this library)
int rt_set_skybox(rt_context *, uint32_t, uint32_t, const float *) { return RT_ERR_NO_DEVICE; }
#include "rt_host_twins.h"
}
