// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a (kernel, device) pair, shared by every host thread and every
// handle: a launch site that sets it to "what this call needs" lowers it under another thread that has set a larger value and
// not launched yet.  Every kernel of this library that may need more than the default 64 KB therefore gets its limit raised
// ONCE per device, to everything the device has (LDS per workgroup minus the kernel's own static __shared__), and it is never
// set again.
#ifndef MSORB_LDS_LIMIT_H
#define MSORB_LDS_LIMIT_H

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

namespace msorb {

// Largest dynamic-LDS size a launch of `fn` may ask for on the CURRENT device, after raising the kernel's limit to it (first
// call per device only).  -1: a query or the raise failed (the caller reports it; the value is not cached, so a later call
// tries again).
inline long long dynamic_lds_room(const void* fn) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, long long> room_of;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(mu);
    auto it = room_of.find({fn, dev});
    if (it != room_of.end()) return it->second;
    int lds_max = 0;
    hipFuncAttributes fa;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess ||
        hipFuncGetAttributes(&fa, fn) != hipSuccess)
        return -1;
    const long long room = (long long)lds_max - (long long)fa.sharedSizeBytes;
    if (room <= 0) return -1;
    if (room > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room) != hipSuccess) return -1;
    room_of[{fn, dev}] = room;
    return room;
}

}  // namespace msorb
#endif
