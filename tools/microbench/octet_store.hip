// Store-pattern microbenchmark: each lane writes 8 consecutive int32 (two 16-byte stores, lane
// stride 32 bytes) at an arbitrary 4-byte alignment, against the plain pattern of one aligned
// 16-byte store per lane.  Prints GB/s written and checks the data.
//   hipcc --offload-arch=gfx950 -O3 -o octet_store octet_store.hip && ./octet_store
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st16(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    u32x4 v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    else *reinterpret_cast<u32x4 *>(p) = v;
}

// MODE 0: lane l writes 4 words at 4 l (one store), twice per 512-word tile (as the quad layout)
// MODE 1: lane l writes 8 words at 8 l (two stores)
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(uint32_t *dst, size_t words_per_wave, int tiles)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t *base = dst + wave * words_per_wave;
    for (int t = 0; t < tiles; ++t) {
        uint32_t *p = base + (size_t)t * 512;
        const uint32_t v0 = (uint32_t)(wave * 131 + t * 7);
        if (MODE == 0) {
            st16<NT>(p + 4 * lane, v0 + 4 * lane, v0 + 4 * lane + 1, v0 + 4 * lane + 2, v0 + 4 * lane + 3);
            st16<NT>(p + 256 + 4 * lane, v0 + 256 + 4 * lane, v0 + 257 + 4 * lane, v0 + 258 + 4 * lane, v0 + 259 + 4 * lane);
        } else {
            st16<NT>(p + 8 * lane, v0 + 8 * lane, v0 + 8 * lane + 1, v0 + 8 * lane + 2, v0 + 8 * lane + 3);
            st16<NT>(p + 8 * lane + 4, v0 + 8 * lane + 4, v0 + 8 * lane + 5, v0 + 8 * lane + 6, v0 + 8 * lane + 7);
        }
    }
}

template <int MODE, bool NT>
static void run(const char *name, uint32_t *d, size_t total_words, int mis)
{
    const int waves = 8192, tiles = 40;
    const size_t wpw = (size_t)tiles * 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemset(d, 0, total_words * 4);
    for (int it = 0; it < 3; ++it) k<MODE, NT><<<waves / 4, 256>>>(d + mis, wpw, tiles);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) k<MODE, NT><<<waves / 4, 256>>>(d + mis, wpw, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(wpw * 2 + 8);
    hipMemcpy(h.data(), d + mis + 5 * wpw, h.size() * 4, hipMemcpyDeviceToHost);
    bool ok = true;
    for (size_t i = 0; i < wpw && ok; ++i) ok = h[i] == (uint32_t)(5 * 131 + (i / 512) * 7 + i % 512);
    printf("%-44s mis=%d  %.1f GB/s  %s\n", name, mis, 10.0 * waves * wpw * 4 / (ms * 1e-3) / 1e9, ok ? "ok" : "WRONG");
}

int main()
{
    const size_t total = (size_t)8192 * 40 * 512 + 64;
    uint32_t *d;
    if (hipMalloc(&d, total * 4) != hipSuccess) return 1;
    for (int mis : {0, 1, 3}) {
        run<0, true>("quads (4 words/lane, 2 stores/tile) nt", d, total, mis == 0 ? 0 : 4 * mis);
        run<1, true>("octets (8 words/lane, stride 32 B) nt", d, total, mis);
        run<1, false>("octets (8 words/lane, stride 32 B) plain", d, total, mis);
    }
    hipFree(d);
    return 0;
}
