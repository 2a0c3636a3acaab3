// Set-up helper of libnanowrap_hip.so: stable radix sort of (Morton key, index) pairs with hipCUB/rocPRIM.
// Runs once per nw_set_points (not on the per-iteration path); kept in its own translation unit because the
// rocPRIM templates dominate the compile time.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

int nw_sort_pairs_u32(const unsigned *key_in, unsigned *key_out, const int *val_in, int *val_out, int n, int bits, hipStream_t stream)
{
    size_t tmp_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key_in, key_out, val_in, val_out, n, 0, bits, stream);
    if (e != hipSuccess) return (int)e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 1);
    if (e != hipSuccess) return (int)e;
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key_in, key_out, val_in, val_out, n, 0, bits, stream);
    hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(tmp);
    return (int)(e != hipSuccess ? e : e2);
}
