// Raising a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process that
// launches on a second GPU must raise it there too.  One LdsLimit per launch site remembers the devices already configured
// (bit per device ordinal, atomic so concurrent host threads agree); devices >= 64 are simply configured on every call.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

struct LdsLimit {
    std::atomic<unsigned long long> done{0};

    // true when every listed kernel may use `bytes` of dynamic LDS on the current device
    template <typename... Fn>
    bool raise(int bytes, Fn... kernels)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
        if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
        const hipError_t errs[] = {hipFuncSetAttribute((const void *)kernels, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)...};
        for (hipError_t e : errs)
            if (e != hipSuccess) return false;
        done.fetch_or(bit, std::memory_order_release);
        return true;
    }
};
