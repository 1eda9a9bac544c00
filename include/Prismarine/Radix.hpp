#pragma once
// Prismarine/Radix.hpp -- psm::RadixSort (reference Include/Prismarine/Radix.hpp:11-79).
// Same public surface; the 8 x (histogram, pfx-work, permute) GL dispatches become one call into
// psm_sort_u64_u32 (prismarine-core_amd/csrc/sort.hip). The reference's 2 Mi key cap is lifted.

#include "Utils.hpp"
#include "Structs.hpp"

namespace NSM {

    class RadixSort {
    public:
        RadixSort() {}
        ~RadixSort() {}

        // stable ascending sort of (u64 key, u32 value) pairs, result in place (Radix.hpp:47-74)
        void sort(GLuint &InKeys, GLuint &InVals, uint32_t size = 1, uint32_t descending = 0) {
            (void)descending;  // never read by the reference's shaders either (radix/includes.glsl:50-55)
            check(psm_sort_u64_u32(context(), InKeys, InVals, size), "RadixSort::sort");
        }
    };
}
