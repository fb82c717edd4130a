// Ceiling of the delta kernel's memory access pattern with no decode work at all:
// NP "pages", one wave each; a wave reads its page's encoded bytes in 1 KiB pieces and writes
// its page's decoded region 1 KiB (64 lanes x 16 B) at a time.
// MODE 0 write only | 1 read (used at once) + write | 2 read prefetched 3 KiB ahead + write
// NT   1 = non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <int MODE, int NT>
__global__ __launch_bounds__(256) void pattern(const u4v *src, u4v *dst, int npages, int out_kib, int in_kib)
{
    const int lane = threadIdx.x & 63;
    const int page = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (page >= npages) return;
    u4v *o = dst + (size_t)page * out_kib * 64;
    const u4v *s = src + (size_t)page * in_kib * 64;
    u4v acc = {(unsigned)page, (unsigned)lane, 0u, 0u};
    int r = 0;
    u4v p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0;
    if (MODE == 2) { p0 = s[lane]; p1 = s[64 + lane]; p2 = s[128 + lane]; r = 3; }
    for (int k = 0; k < out_kib; ++k) {
        if (MODE == 1 && (k * in_kib) / out_kib >= r && r < in_kib) {
            u4v v = s[r * 64 + lane];
            acc ^= v;
            ++r;
        }
        if (MODE == 2 && (k % 12) == 11) {   // every 12 KiB out ~ 3 KiB in
            acc ^= p0; acc += p1; acc ^= p2;
            if (r + 3 <= in_kib) { p0 = s[r * 64 + lane]; p1 = s[(r + 1) * 64 + lane]; p2 = s[(r + 2) * 64 + lane]; r += 3; }
        }
        if (NT) __builtin_nontemporal_store(acc, &o[k * 64 + lane]); else o[k * 64 + lane] = acc;
        acc.x += 1;
    }
}

template <int MODE, int NT>
static void run(const u4v *src, u4v *dst, int npages, int out_kib, int in_kib, size_t ob, size_t ib)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        CHECK(hipEventRecord(a));
        for (int k = 0; k < 10; ++k)
            hipLaunchKernelGGL((pattern<MODE, NT>), dim3((npages + 3) / 4), dim3(256), 0, 0, src, dst, npages, out_kib, in_kib);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
        if (ms < best) best = ms;
    }
    double bytes = MODE == 0 ? (double)ob : (double)ob + ib;
    printf("mode %d nt %d: %.4f ms  %.0f GB/s total, %.0f GB/s written\n", MODE, NT, best, bytes / best / 1e6, ob / best / 1e6);
}

int main(int argc, char **argv)
{
    int npages = argc > 1 ? atoi(argv[1]) : 5000;
    int out_kib = 157, in_kib = 40;
    size_t ob = (size_t)npages * out_kib * 1024, ib = (size_t)npages * in_kib * 1024;
    u4v *src, *dst;
    CHECK(hipMalloc(&src, ib)); CHECK(hipMalloc(&dst, ob));
    CHECK(hipMemset(src, 1, ib)); CHECK(hipMemset(dst, 0, ob));
    run<0, 0>(src, dst, npages, out_kib, in_kib, ob, ib);
    run<0, 1>(src, dst, npages, out_kib, in_kib, ob, ib);
    run<1, 0>(src, dst, npages, out_kib, in_kib, ob, ib);
    run<2, 0>(src, dst, npages, out_kib, in_kib, ob, ib);
    run<2, 1>(src, dst, npages, out_kib, in_kib, ob, ib);
    return 0;
}
