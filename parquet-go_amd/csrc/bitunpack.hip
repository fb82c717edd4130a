// bitunpack.hip — fixed-width bit-unpack of a contiguous run of 8-value groups.
//
// Semantics follow the reference's unpack8int32_w / unpack8int64_w tables
// (encoding/bitpacking32.go:10-44,81-1154; encoding/bitpacking64.go:10-76,145-3314):
// group g is the w bytes at src + g*w and value k of it is bits [k*w,(k+1)*w) of
// that little-endian group.  Because every group is exactly w bytes, the groups
// concatenate into one LSB-first bit stream in which value i sits at bit i*w, so
// the kernel is a pure streaming transform:
//   HBM --16 B/lane coalesced--> LDS tile --2..3 dword reads + funnel shift--> regs
//   --16 B/lane coalesced--> HBM.
// One workgroup = one tile of 4096 int32 / 2048 int64 values (<= 16 KiB encoded).
#include "pq_device.hpp"
#include "pq_launch.hpp"

namespace pq {

constexpr int kTileBytesMax = 16384;

template <typename T> struct UnpackCfg;
template <> struct UnpackCfg<int32_t> {
    static constexpr int VPT = 4;             // values per thread per step (16 B)
    static constexpr int TILE = 4096;         // values per workgroup
};
template <> struct UnpackCfg<int64_t> {
    static constexpr int VPT = 2;
    static constexpr int TILE = 2048;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(kWG) void bitunpack_kernel(const uint8_t *__restrict__ src,
                                                        uint64_t src_len, uint32_t w,
                                                        uint64_t n_values, T *__restrict__ dst)
{
    using Cfg = UnpackCfg<T>;
    constexpr int VPT = Cfg::VPT;
    constexpr int TILE = Cfg::TILE;
    __shared__ __attribute__((aligned(16))) uint32_t lds[(kTileBytesMax + 32) / 4 + 4];

    const int tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint64_t v0 = tile * (uint64_t)TILE;
    const uint64_t tile_bytes = (uint64_t)(TILE / 8) * w;
    const uint8_t *S = src + tile * tile_bytes;
    const int64_t len = (int64_t)min(tile_bytes, src_len - min(src_len, tile * tile_bytes));
    const int64_t c0 = chunk_base(S, 0); // -15..0

    if (w != 0) {
        uint4 *l4 = reinterpret_cast<uint4 *>(lds);
        const int pieces = (int)((len - c0 + 15) >> 4) + 1; // one spare piece of zeros
        for (int i = tid; i < pieces; i += kWG) {
            int64_t o = c0 + 16 * (int64_t)i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o + 16 > 0 && o < len) {
                v = *reinterpret_cast<const uint4 *>(S + o);
                if (o < 0 || o + 16 > len) v = mask_piece(v, -o, len - o);
            }
            l4[i] = v;
        }
        __syncthreads();
    }

    const uint32_t phase_bits = (uint32_t)(-c0) * 8u;
#pragma unroll
    for (int it = 0; it < TILE / (kWG * VPT); ++it) {
        const uint32_t v = (uint32_t)(it * kWG * VPT + tid * VPT);
        const uint64_t gi = v0 + v;
        if (gi >= n_values) break; // n_values is a multiple of 8, VPT divides 8
        T out[VPT];
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < VPT; ++j) out[j] = 0;
        } else {
            const uint32_t bit = phase_bits + v * w;
#pragma unroll
            for (int j = 0; j < VPT; ++j) {
                if constexpr (sizeof(T) == 4)
                    out[j] = (T)extract32(lds, bit + j * w, w);
                else
                    out[j] = (T)extract64(lds, bit + j * w, w);
            }
        }
        if constexpr (VEC) {
            uint4 pk;
            if constexpr (sizeof(T) == 4) {
                pk = make_uint4((uint32_t)out[0], (uint32_t)out[1], (uint32_t)out[2], (uint32_t)out[3]);
            } else {
                pk = make_uint4((uint32_t)(uint64_t)out[0], (uint32_t)((uint64_t)out[0] >> 32),
                                (uint32_t)(uint64_t)out[1], (uint32_t)((uint64_t)out[1] >> 32));
            }
            store_nt16(dst + gi, pk.x, pk.y, pk.z, pk.w);  // written once: past the caches, which keep the input
        } else {
#pragma unroll
            for (int j = 0; j < VPT; ++j) dst[gi + j] = out[j];
        }
    }
}

template <typename T>
static hipError_t launch_bitunpack_t(hipStream_t stream, const void *d_src, size_t src_len, int w,
                                     size_t n_groups, T *d_dst)
{
    if (n_groups == 0) return hipSuccess;
    const uint64_t n_values = (uint64_t)n_groups * 8u;
    const uint64_t tiles = (n_values + UnpackCfg<T>::TILE - 1) / UnpackCfg<T>::TILE;
    if (tiles > 0x7fffffffull) return hipErrorInvalidValue;
    const bool vec = (reinterpret_cast<uintptr_t>(d_dst) & 15u) == 0;
    dim3 grid((unsigned)tiles), block(kWG);
    if (vec)
        hipLaunchKernelGGL((bitunpack_kernel<T, true>), grid, block, 0, stream,
                           (const uint8_t *)d_src, (uint64_t)src_len, (uint32_t)w, n_values, d_dst);
    else
        hipLaunchKernelGGL((bitunpack_kernel<T, false>), grid, block, 0, stream,
                           (const uint8_t *)d_src, (uint64_t)src_len, (uint32_t)w, n_values, d_dst);
    return hipGetLastError();
}

hipError_t launch_bitunpack32(hipStream_t s, const void *src, size_t len, int w, size_t ng, int32_t *dst)
{
    return launch_bitunpack_t<int32_t>(s, src, len, w, ng, dst);
}

hipError_t launch_bitunpack64(hipStream_t s, const void *src, size_t len, int w, size_t ng, int64_t *dst)
{
    return launch_bitunpack_t<int64_t>(s, src, len, w, ng, dst);
}

} // namespace pq
