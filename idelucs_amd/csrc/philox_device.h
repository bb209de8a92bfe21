// Philox4x32-10 as the training step's dropout streams use it (train_step.hip, l1_fwd.hip): one definition, so that a kernel that
// takes over another kernel's activation keeps its masks bit for bit.  Counter words / keys are the caller's.
#pragma once
#include <stdint.h>

namespace idl_dev {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

}  // namespace idl_dev
