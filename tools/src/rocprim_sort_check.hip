// rocprim::radix_sort_keys(double_buffer) on 64-bit keys over a bit range, for sizes in its merge-sort range: is the result sorted?
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/rocprim_sort_check tools/src/rocprim_sort_check.hip
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const unsigned sizes[] = {1000, 1024, 1025, 2048, 4032, 4096, 4097, 4425, 4983, 7933, 7960, 8192, 8193, 27466, 28364, 52640, 65536, 100000, 224998, 1048576, 1048577, 1200000, 3000000};
    for (unsigned mode = 0; mode < 5; mode++) {
        const unsigned bits_lo = mode == 0 ? 0u : mode == 1 ? 13u : mode == 2 ? 16u : 0u, bits_hi = mode == 3 ? 63u : mode == 4 ? 51u : 64u;
        for (unsigned n : sizes) {
            std::vector<unsigned long long> h(n);
            unsigned long long s = 88172645463325252ull + n;
            for (unsigned i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (s & ~((1ull << 16) - 1ull)) | i; if (bits_hi < 64u) h[i] &= (1ull << bits_hi) - 1ull; }
            unsigned long long *a, *b; void *tmp = nullptr; size_t tb = 0;
            CK(hipMalloc(&a, n * 8ull)); CK(hipMalloc(&b, n * 8ull));
            CK(hipMemcpy(a, h.data(), n * 8ull, hipMemcpyHostToDevice));
            rocprim::double_buffer<unsigned long long> d(a, b);
            CK(rocprim::radix_sort_keys(nullptr, tb, d, n, bits_lo, bits_hi, 0));
            // the library's callers often hand over a larger scratch buffer than asked for
            for (size_t extra : {(size_t)0, (size_t)1 << 20}) {
                CK(hipMemcpy(a, h.data(), n * 8ull, hipMemcpyHostToDevice));
                rocprim::double_buffer<unsigned long long> dd(a, b);
                size_t have = tb + extra;
                CK(hipMalloc(&tmp, have));
                CK(rocprim::radix_sort_keys(tmp, have, dd, n, bits_lo, bits_hi, 0));
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> r(n);
                CK(hipMemcpy(r.data(), dd.current(), n * 8ull, hipMemcpyDeviceToHost));
                unsigned bad = 0, first = 0;
                for (unsigned i = 1; i < n; i++) if ((r[i - 1] >> bits_lo) > (r[i] >> bits_lo)) { if (!bad) first = i; bad++; }
                printf("bits [%2u,%2u) n %7u scratch %8zu (+%zu): %s", bits_lo, bits_hi, n, tb, extra, bad ? "NOT SORTED" : "sorted");
                if (bad) printf(" (%u descents, first at %u)", bad, first);
                printf("\n");
                CK(hipFree(tmp));
            }
            CK(hipFree(a)); CK(hipFree(b));
        }
    }
    // (code, index) PAIRS over bits [0, 63): the static builds' sort
    for (unsigned n : sizes) {
        std::vector<unsigned long long> h(n); std::vector<unsigned> hv(n);
        unsigned long long s = 1234567ull + n;
        for (unsigned i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (s >> 1) >> (i % 3 ? 0 : 40); hv[i] = i; }      // a third of the keys collide in 23 bits
        unsigned long long *a, *b; unsigned *va, *vb; void *tmp = nullptr; size_t tb = 0;
        CK(hipMalloc(&a, n * 8ull)); CK(hipMalloc(&b, n * 8ull)); CK(hipMalloc(&va, n * 4ull)); CK(hipMalloc(&vb, n * 4ull));
        CK(hipMemcpy(a, h.data(), n * 8ull, hipMemcpyHostToDevice)); CK(hipMemcpy(va, hv.data(), n * 4ull, hipMemcpyHostToDevice));
        rocprim::double_buffer<unsigned long long> d(a, b); rocprim::double_buffer<unsigned> dv(va, vb);
        CK(rocprim::radix_sort_pairs(nullptr, tb, d, dv, n, 0u, 63u, 0));
        CK(hipMalloc(&tmp, tb));
        CK(rocprim::radix_sort_pairs(tmp, tb, d, dv, n, 0u, 63u, 0));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> r(n); std::vector<unsigned> rv(n);
        CK(hipMemcpy(r.data(), d.current(), n * 8ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(rv.data(), dv.current(), n * 4ull, hipMemcpyDeviceToHost));
        unsigned bad = 0;
        for (unsigned i = 0; i < n; i++) {
            if (rv[i] >= n || h[rv[i]] != r[i]) bad++;                                   // the value travelled with its key
            if (i && (r[i - 1] > r[i] || (r[i - 1] == r[i] && rv[i - 1] > rv[i]))) bad++;   // sorted, and stable
        }
        printf("pairs [ 0,63) n %7u: %s\n", n, bad ? "NOT SORTED / NOT STABLE" : "sorted, stable");
        CK(hipFree(tmp)); CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(va)); CK(hipFree(vb));
    }
    return 0;
}
