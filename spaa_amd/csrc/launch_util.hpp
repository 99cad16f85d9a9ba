// launch_util.hpp — host-side helpers shared by the launchers.
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr int SPAA_MAX_DEVICES = 32;

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: a process that drives several GPUs
// (one AttackState per device) must set it on each of them, once.  `done` is the calling launcher's own static table.
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, bool (&done)[SPAA_MAX_DEVICES]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= SPAA_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    return hipSuccess;
}

}  // namespace
