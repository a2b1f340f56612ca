// util/context.hpp -- device selection handle.
//
// The reference threads a moderngpu `CudaContext&` through every Enact() call
// (gunrock/app/bfs/bfs_enactor.cuh:573-579; created by mgpu::CreateCudaDevice, bfs_app.cu:392) because
// its advance operator borrows moderngpu's temp allocator and streams.  Nothing here needs a
// third-party allocator, so the context shrinks to "which GPU" -- kept so call sites read the same.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace util {

struct DeviceContext {
    int device = 0;
    hipError_t status = hipSuccess;
    explicit DeviceContext(int dev = 0) : device(dev)
    {
        status = GRError(hipSetDevice(dev), "DeviceContext hipSetDevice failed", __FILE__, __LINE__);
    }
};

}  // namespace util
}  // namespace gunrock
